"""Synthetic datasets for the approximate-search legs of bench.py, their hardness statistics, and the graph index's request path in
the reference's call shape.  Imported by bench.py (and scripts/hardness_probe.py); everything runs on the device through torch
(synthetic rows) and the C ABI (searches).

Three sets, all unit-norm fp16 rows of width 1152:
  easy   the round-1..4 set: rows/50 centres around rows/5000 super-centres, noise 0.3 -- the true top-10 of a query are its ~50
         cluster-mates; a beam search reaches recall 0.98 with a search list of 12
  hard   no micro-clusters: a common mean direction (the "cone" of contrastive embeddings: random pairs have cosine ~ 0.45), a
         low-rank Gaussian with a power-law spectrum (rank 96) around power-law sized topic centres, isotropic noise on top.
         Neighbours are separated from non-neighbours by a margin of a few hundredths of cosine, not by 0.3
  ood    the hard base set queried from ANOTHER distribution, as text queries against image embeddings are
         (src/generate_index_shard.rs:62-84,127-131): queries = the hard mixture with its own topic mass, pushed along a fixed
         "modality gap" direction and with more isotropic noise; the graph is built with a query sample appended after the base rows
         (query_breakpoint) and robust_stitch (diskann/src/lib.rs:326-374), as the reference's OOD-DiskANN variant does
"""
import math
import time

D = 1152


def easy_generator(n):
    """rows/50 centres around rows/5000 super-centres, noise 0.3 -> f(m, seed) returning an [m, 1152] fp16 device tensor."""
    import torch
    g0 = torch.Generator(device="cuda").manual_seed(0)
    hier = max(8, n // 5000)
    sup = torch.randn(hier, D, device="cuda", generator=g0)
    sup /= sup.norm(dim=1, keepdim=True)
    nc_ = max(64, n // 50)
    centres = sup[torch.randint(0, hier, (nc_,), device="cuda", generator=g0)] + torch.randn(nc_, D, device="cuda", generator=g0) * (0.7 / D ** 0.5)
    centres /= centres.norm(dim=1, keepdim=True)

    def clustered(m, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        out = torch.empty(m, D, device="cuda", dtype=torch.float16)
        for i in range(0, m, 1 << 18):
            c = min(1 << 18, m - i)
            x = centres[torch.randint(0, nc_, (c,), device="cuda", generator=g)] + torch.randn(c, D, device="cuda", generator=g) * (0.3 / D ** 0.5)
            out[i:i + c] = (x / x.norm(dim=1, keepdim=True)).half()
        return out

    return clustered


class HardSet:
    """normalise(cone * mu + topic centre + within-topic low-rank Gaussian + isotropic noise).  Parameters are fractions of the
    squared norm before normalisation: cone^2 + topic^2 + within^2 + noise^2 = 1."""

    def __init__(self, n, rank=96, n_topics=None, cone=0.62, topic=0.45, within=0.55, noise=0.33, decay=0.6, zipf=1.0, seed=0):
        import torch
        g0 = torch.Generator(device="cuda").manual_seed(1000 + seed)
        self.rank = rank
        self.n_topics = n_topics or max(64, int(round(n ** 0.5 / 2)))
        s = math.sqrt(cone ** 2 + topic ** 2 + within ** 2 + noise ** 2)
        self.cone, self.topic, self.within, self.noise = cone / s, topic / s, within / s, noise / s
        q, _ = torch.linalg.qr(torch.randn(D, rank + 2, device="cuda", generator=g0))
        self.mu = q[:, 0].contiguous()                       # the cone axis
        self.gap = q[:, 1].contiguous()                      # the modality-gap direction of the OOD queries (orthogonal to everything else)
        basis = q[:, 2:].contiguous()                        # [D, rank] orthonormal
        spec = torch.arange(1, rank + 1, device="cuda", dtype=torch.float32) ** (-decay)
        spec /= spec.norm()                                  # power-law spectrum, unit total energy
        self.A = (basis * spec[None, :]).contiguous()        # z ~ N(0, I_rank) -> A z has unit expected squared norm
        z = torch.randn(self.n_topics, rank, device="cuda", generator=g0)
        self.centres = z @ self.A.T                          # topic centres live in the low-rank subspace
        w = torch.arange(1, self.n_topics + 1, device="cuda", dtype=torch.float32) ** (-zipf)
        self.topic_p = w / w.sum()                           # power-law topic sizes
        wq = w[torch.randperm(self.n_topics, device="cuda", generator=g0)]
        self.topic_p_queries = wq / wq.sum()                 # OOD queries: another topic mass over the same topics

    def rows(self, m, seed, queries=None, gap=0.0, extra_noise=0.0):
        """[m, 1152] fp16 device tensor.  queries='ood': topic mass of the query distribution, pushed `gap` along the gap direction,
        `extra_noise` more isotropic noise."""
        import torch
        g = torch.Generator(device="cuda").manual_seed(seed)
        out = torch.empty(m, D, device="cuda", dtype=torch.float16)
        p = self.topic_p_queries if queries == "ood" else self.topic_p
        noise = math.sqrt(self.noise ** 2 + extra_noise ** 2)
        for i in range(0, m, 1 << 18):
            c = min(1 << 18, m - i)
            t = torch.multinomial(p, c, replacement=True, generator=g)
            z = torch.randn(c, self.rank, device="cuda", generator=g)
            x = self.cone * self.mu[None, :] + self.topic * self.centres[t] + self.within * (z @ self.A.T)
            x += torch.randn(c, D, device="cuda", generator=g) * (noise / D ** 0.5)
            if gap:
                x += gap * self.gap[None, :]
            out[i:i + c] = (x / x.norm(dim=1, keepdim=True)).half()
        return out


def hardness(vecs, searcher, rows, queries_f16, k=10, k_lid=20, n_random=4096):
    """Statistics of a (base, query) pair that say how hard approximate search is, from exact brute force on `queries_f16`:
      relative_contrast   mean dot of a random base row with the query / ... reported as both means and the classic ratio in DISTANCE
                          form: mean distance to a random row / mean distance to the k-th neighbour (distance = 1 - dot; unit rows)
      lid_mle             Levina-Bickel / Amsaleg MLE of the local intrinsic dimension at k_lid neighbours, on Euclidean distances
                          sqrt(2 - 2 dot), averaged over the queries
    Returns (stats dict, exact top-k ids)."""
    import numpy as np
    import torch
    import mse
    q = queries_f16
    nq = q.shape[0]
    sc, ids = searcher.bruteforce_topk(q.cpu().numpy().view(np.uint16), max(k, k_lid))
    dots = sc.astype(np.float64) / float(mse.SCALE)
    sel = torch.randint(0, rows.shape[0], (n_random,), device="cuda")
    rnd = (q.float() @ rows[sel].float().T).cpu().numpy().astype(np.float64)        # [nq, n_random]
    d_k = 1.0 - dots[:, k - 1]
    d_rand = 1.0 - rnd.mean(axis=1)
    r = np.sqrt(np.maximum(2.0 - 2.0 * dots[:, :k_lid], 1e-12))                     # ascending distances
    lid = -1.0 / np.mean(np.log(np.maximum(r[:, :-1], 1e-12) / r[:, -1:]), axis=1)
    return ({"queries": int(nq), "mean_dot_nn1": float(dots[:, 0].mean()), f"mean_dot_nn{k}": float(dots[:, k - 1].mean()),
             "mean_dot_random_row": float(rnd.mean()), "relative_contrast_at_%d" % k: float(d_rand.mean() / d_k.mean()),
             "lid_mle_k%d" % k_lid: float(np.median(lid)), "note": "relative contrast = mean (1 - dot) to a random row / mean (1 - dot) to the %d-th neighbour; "
             "LID = MLE over %d neighbours on Euclidean distances, median over queries" % (k, k_lid)}, ids[:, :k])


def recall_at(top, truth):
    k = truth.shape[1]
    return sum(len(set(top[i, :k].tolist()) & set(truth[i].tolist())) for i in range(truth.shape[0])) / (k * truth.shape[0])
