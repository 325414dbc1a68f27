"""mse: host-side mirror of the reference's scoring interfaces over libmse_hip.so (MI355X only).

No CPU fallback exists: importing is cheap, but any compute call raises MseError when the HIP
library is not built or no device is present."""
from .ffi import MseError, LIB_PATH  # noqa: F401
from .vector import (SCALE, ID_NONE, MODE_AUTO, MODE_EXACT, MODE_MFMA, scale_dot_result, scale_dot_result_f64,  # noqa: F401
                     fast_dot, fast_dot_noprefetch, VectorList, Searcher, Dispatcher, ProductQuantizer, QueryLUT, Codes,
                     descriptor_product)
from .diskann import (NeighbourBuffer, IndexGraph, greedy_search, disk_greedy_search, DiskSearchResult, medioid,  # noqa: F401
                      select_shard, dedup_visited, DUPLICATES_THRESHOLD, DeviceGraph, disk_search_batch, IndexBuildConfig,
                      BuildGraph, robust_prune, topk_of_visited, set_entries, disk_query_topk, QueryTickets, set_entry_centroids,
                      set_coalescer, coalescer_stats, set_dedup)
from .index import ScalarQuantizerIndex  # noqa: F401
from .common import decode_fp16_buffer, chunk_fp16_buffer, get_total_embedding  # noqa: F401
from .index_pack import ScoreModel, descriptor_buckets  # noqa: F401
from .shard import ShardGroup, Comm, shard_range  # noqa: F401
