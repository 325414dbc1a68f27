"""Embedding server with the wire contract of the reference's clip_server.py, backed by the HIP engine.

Contract (reference clip_server.py) -- kept byte compatible:
  POST /        body = msgpack map {"images": [bytes, ...]} or {"text": [str, ...]}          (:151-170)
                200 -> msgpack array of bin, each `embedding_size * 2` bytes little-endian fp16,
                       L2-normalised                                                          (:115,166)
                500 -> msgpack string with the error text; batch > max_batch_size and a body with
                       neither key are errors                                                  (:136-146,167-170)
                a full request queue raises queue.Full out of the handler (put_nowait, :161)
  GET  /config  msgpack map {"model", "batch", "image_size": [w, h], "embedding_size"}        (:176-183)
  GET  /        204                                                                            (:185-187)
  GET  /metrics Prometheus text: modelserver_total_items{model,modality},
                modelserver_inftime{model,batch_size}, modelserver_batchcount{model}           (:86-88,189-191)
  request bodies up to 64 MiB                                                                  (:148)
Stages as in the reference: aiohttp handler -> preprocessing thread -> model thread, two bounded queues of 10
(:125,130); a request travels as a `Job` whose asyncio future the handler awaits.  Images that are what the reference's
clients send -- 24-bit BMP of the model's size (src/common.rs:31-54) -- skip the host decoder: the request bytes go to the
device as they are (SiglipImageEngine.encode_bmp); anything else is decoded with PIL on the preprocessing thread.  Configuration = JSON file given as argv[1] with the reference's keys
(`device`, `model`, `model_path`, `model_name`, `max_batch_size`, `port`).

The model call is the seam the reference fills with `fast_image_fns` / `model.encode_image`
(:66-82,105-114): here `engine.encode_image(images fp16 NCHW) -> [b, emb] f32`.  The engine is injected, so
the contract is testable without a GPU (tests/test_clip_server.py uses a stand-in engine).
"""
import asyncio
import io
import json
import queue
import sys
import threading
import traceback

import msgpack
import numpy as np

def preprocess_image(data: bytes, size):
    """open_clip's preprocess for ViT-SO400M-14-SigLIP-384 as the reference relies on it:
    decode, RGB, resize to (w, h) if needed (bicubic, squash), ToTensor, Normalize(mean=std=0.5), .half().
    Clients already send 384x384 24-bit BMP (src/common.rs:50-53), so the resize is normally the identity."""
    a = decode_image(data, size).astype(np.float32)          # [h, w, 3]
    a = a / np.float32(127.5) - np.float32(1.0)               # (x/255 - 0.5) / 0.5
    return np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float16)


def decode_image(data: bytes, size):
    """Host decoder for everything that is not the clients' own format: decode, RGB, resize if needed -> uint8 [h, w, 3].
    The arithmetic half (x/127.5 - 1, fp16, NCHW) then runs on the device (SiglipImageEngine.encode_rgb8)."""
    from PIL import Image
    im = Image.open(io.BytesIO(data)).convert("RGB")
    if im.size != tuple(size):
        im = im.resize(tuple(size), Image.BICUBIC)
    return np.asarray(im, dtype=np.uint8)


class Job:
    """One POST on its way through the two worker threads.  The handler awaits `done`; whichever stage ends the job --
    a failed check, a failed decode, the model -- resolves it from its own thread."""

    __slots__ = ("text", "images", "done", "_loop", "stage")

    def __init__(self, text, images, loop=None):
        self.text, self.images = text, images
        self._loop = loop
        self.done = loop.create_future() if loop is not None else None
        self.stage = None          # what the preprocessing step hands to the model: ("tokens" | "bmp" | "rgb8" | "nchw", payload)

    def finish(self, ok, payload):
        if self.done is not None:
            self._loop.call_soon_threadsafe(self._resolve, ok, payload)

    def _resolve(self, ok, payload):
        if not self.done.done():
            self.done.set_result((ok, payload))


class ClipServer:
    QUEUE_DEPTH = 10           # both hand-off queues of the reference hold 10 (clip_server.py:125,130)

    def __init__(self, config, image_engine, text_engine=None, tokenizer=None, registry=None):
        from prometheus_client import CollectorRegistry, Counter, Histogram
        self.config = config
        self.bs = int(config["max_batch_size"])
        self.model_name = config["model_name"]
        # one engine, or several replicas of it: every replica gets its own model thread, so the host side of one batch (queue
        # hand-off, upload of the raw files, download of the rows) runs beside the device side of another
        self.image_engines = list(image_engine) if isinstance(image_engine, (list, tuple)) else [image_engine]
        image_engine = self.image_engines[0]
        self.image_engine = image_engine
        self.text_engine = text_engine
        self.tokenizer = tokenizer
        self.image_size = tuple(getattr(image_engine, "image_size", (384, 384)))
        self.embedding_size = int(image_engine.embedding_size)
        self.registry = registry or CollectorRegistry()
        self.items_ctr = Counter("modelserver_total_items", "Items run through model server", ["model", "modality"],
                                 registry=self.registry)
        self.inference_time_hist = Histogram("modelserver_inftime", "Time running inference", ["model", "batch_size"],
                                             registry=self.registry)
        self.batch_count_ctr = Counter("modelserver_batchcount", "Inference batches run", ["model"], registry=self.registry)
        self.prep_q = queue.Queue(self.QUEUE_DEPTH)     # handler -> preprocessing
        self.model_q = queue.Queue(self.QUEUE_DEPTH)    # preprocessing -> model
        self._threads = []
        self._stop = object()

    def submit(self, job):
        """What the POST handler does with a request: a full queue raises queue.Full out of the handler, as the reference's
        put_nowait does (clip_server.py:161)."""
        self.prep_q.put_nowait(job)

    # ---- preprocessing stage (the reference's preprocessing_thread, clip_server.py:131-146) ----
    def prepare(self, job):
        if job.text:
            assert len(job.text) <= self.bs, f"max batch size is {self.bs}"
            if self.tokenizer is None:
                raise RuntimeError("tokenizer not available")
            return "tokens", np.asarray(self.tokenizer(job.text))
        if job.images:
            assert len(job.images) <= self.bs, f"max batch size is {self.bs}"
            eng = self.image_engine
            if hasattr(eng, "encode_bmp"):
                from .siglip import is_plain_bmp
                if all(is_plain_bmp(im, self.image_size) for im in job.images):
                    return "bmp", list(job.images)          # the clients' own format: no host decode at all
            if hasattr(eng, "encode_rgb8"):
                return "rgb8", np.stack([decode_image(im, self.image_size) for im in job.images])
            return "nchw", np.stack([preprocess_image(im, self.image_size) for im in job.images])
        raise AssertionError("images or text required")

    # ---- model stage (do_inference, clip_server.py:91-123) ----
    def run_model(self, kind, payload, engine=None):
        engine = engine or self.image_engine
        n = len(payload)
        if kind == "tokens":
            if self.text_engine is None:
                raise RuntimeError("text tower not loaded")
            self.items_ctr.labels(self.model_name, "text").inc(n)
            with self.inference_time_hist.labels(self.model_name + "-text", n).time():
                f = np.asarray(self.text_engine.encode_text(payload), np.float32)
                f = f / np.linalg.norm(f, axis=-1, keepdims=True)
        else:
            self.items_ctr.labels(self.model_name, "image").inc(n)
            with self.inference_time_hist.labels(self.model_name + "-image", n).time():
                # the engine normalises on the device; result rows are unit norm like `features /= norm`
                call = {"bmp": "encode_bmp", "rgb8": "encode_rgb8", "nchw": "encode_image"}[kind]
                f = np.asarray(getattr(engine, call)(payload), np.float32)
        self.batch_count_ctr.labels(self.model_name).inc()
        return f

    def _stage_loop(self, inbox, work):
        while True:
            job = inbox.get()
            if job is self._stop:
                return
            try:
                work(job)
            except Exception as e:  # noqa: BLE001 - every failure is reported to the client as a 500 string
                traceback.print_exc()
                job.finish(False, str(e))

    def _prep_work(self, job):
        job.stage = self.prepare(job)
        self.model_q.put(job)

    def _model_work(self, job, engine=None):
        job.finish(True, self.run_model(*job.stage, engine=engine))

    def _model_loop(self, engine=None):
        """The model thread (do_inference's loop, clip_server.py:91-123,126-128) with one addition the wire contract does not see:
        BMP jobs already waiting in the queue are run as ONE engine call, up to the engine's own batch capacity (a request is
        limited to max_batch_size images and 64 MiB; the tower is most efficient at 256).  Rows go back to their own requests."""
        engine = engine or self.image_engine
        # never more than the engine takes in one call; a request larger than that fails on its own, as in the reference (:139)
        cap = int(getattr(engine, "max_batch", self.bs))
        held = None
        while True:
            if held is not None:
                job, held = held, None
            else:
                job = self.model_q.get()
            if job is self._stop:
                self._hand_on_stop()           # the other model threads stop on it as well
                return
            group = [job]
            try:
                kind, payload = job.stage
                if kind == "bmp":
                    total = len(payload)
                    while total < cap:
                        try:
                            nxt = self.model_q.get_nowait()
                        except queue.Empty:
                            break
                        if nxt is self._stop or nxt.stage[0] != "bmp" or total + len(nxt.stage[1]) > cap:
                            held = nxt             # runs next, on its own
                            break
                        group.append(nxt)
                        total += len(nxt.stage[1])
                if len(group) == 1:
                    self._model_work(job, engine)
                    group = []
                    continue
                try:
                    rows = self.run_model("bmp", [im for j in group for im in j.stage[1]], engine)
                except Exception:  # noqa: BLE001
                    # the shared call failed: every request of the group is run again on its own, so that a request only ever
                    # fails for what IT sent (the reference fails the offending request alone)
                    traceback.print_exc()
                    pending, group = group, []
                    for j in pending:
                        try:
                            self._model_work(j, engine)
                        except Exception as e1:  # noqa: BLE001
                            j.finish(False, str(e1))
                    continue
                at = 0
                while group:
                    j = group[0]
                    n = len(j.stage[1])
                    j.finish(True, rows[at:at + n])
                    group.pop(0)           # answered: must not be answered again if a later hand-back fails
                    at += n
            except Exception as e:  # noqa: BLE001 - every failure is reported to the client as a 500 string
                traceback.print_exc()
                for j in group:
                    j.finish(False, str(e))

    def _hand_on_stop(self):
        """Pass the stop marker to the next model thread without ever blocking on a full queue: jobs still queued are answered
        with an error first (nobody may wait for ever on a server that is going down)."""
        while True:
            try:
                self.model_q.put_nowait(self._stop)
                return
            except queue.Full:
                try:
                    j = self.model_q.get_nowait()
                except queue.Empty:
                    continue
                if j is not self._stop:
                    j.finish(False, "server is shutting down")

    def start_threads(self):
        for eng in self.image_engines:
            th = threading.Thread(target=self._model_loop, args=(eng,), daemon=True)
            th.start()
            self._threads.append(th)
        th = threading.Thread(target=self._stage_loop, args=(self.prep_q, self._prep_work), daemon=True)
        th.start()
        self._threads.append(th)

    def stop_threads(self):
        for q_ in (self.prep_q, self.model_q):
            while True:
                try:
                    q_.put_nowait(self._stop)
                    break
                except queue.Full:         # make room: a job that can no longer be served is told so
                    try:
                        j = q_.get_nowait()
                    except queue.Empty:
                        continue
                    if j is not self._stop:
                        j.finish(False, "server is shutting down")

    # ---- HTTP (clip_server.py:148-200) ----
    def make_app(self):
        from aiohttp import web
        from prometheus_client import generate_latest
        msgpack_type = "application/msgpack"

        async def embed(request):
            data = await request.read()
            loop = asyncio.get_running_loop()
            # a 128-image request is 56 MB of msgpack: decoding it on the event loop (as the reference does, :150) holds up every
            # other connection for tens of milliseconds; large bodies are decoded on the default executor instead
            body = msgpack.loads(data) if len(data) < (1 << 20) else await loop.run_in_executor(None, msgpack.loads, data)
            job = Job(body.get("text"), body.get("images"), loop)
            self.submit(job)
            ok, payload = await job.done
            if ok:
                payload = [row.astype("float16").tobytes() for row in payload]
            return web.Response(body=msgpack.dumps(payload), status=200 if ok else 500, content_type=msgpack_type)

        async def describe(request):
            info = {"model": self.config["model"], "batch": self.bs, "image_size": self.image_size, "embedding_size": self.embedding_size}
            return web.Response(body=msgpack.dumps(info), status=200, content_type=msgpack_type)

        async def alive(request):
            return web.Response(status=204)

        async def metrics(request):
            return web.Response(body=generate_latest(self.registry))

        app = web.Application(client_max_size=64 << 20)
        app.router.add_post("/", embed)
        app.router.add_get("/config", describe)
        app.router.add_get("/", alive)
        app.router.add_get("/metrics", metrics)
        return app


class SiglipTokenizer:
    """`open_clip.get_tokenizer("ViT-SO400M-14-SigLIP-384")` as clip_server.py:100,129 uses it: text is
    canonicalised (lower case, punctuation removed, whitespace collapsed), encoded with the c4_en sentencepiece
    model, an end marker equal to the pad id (1) is appended, and rows are truncated / padded to the context
    length.  The sentencepiece model file is not redistributable from here: `tokenizer_path` in the config."""

    def __init__(self, model_path=None, context_length=64, pad_id=1, model_proto=None):
        import sentencepiece
        self.sp = (sentencepiece.SentencePieceProcessor(model_proto=model_proto) if model_proto is not None
                   else sentencepiece.SentencePieceProcessor(model_file=model_path))
        self.context_length = context_length
        self.pad_id = pad_id

    @staticmethod
    def canonicalize(text):
        import string
        text = text.translate(str.maketrans("", "", string.punctuation)).lower()
        return " ".join(text.split())

    def __call__(self, texts):
        from .siglip import pad_tokens
        if isinstance(texts, str):
            texts = [texts]
        rows = []
        for t in texts:
            ids = self.sp.encode(self.canonicalize(t))[: self.context_length - 1]
            rows.append(ids + [self.pad_id])
        return pad_tokens(rows, self.context_length, self.pad_id)


def _load_state(config):
    path = config.get("model_path")
    if not path:
        return None
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file
        return load_file(path)
    import torch
    state = torch.load(path, map_location="cpu")
    return state.get("state_dict", state)


def load_engine(config, state=None):
    """Build the HIP image engine from the reference's config keys.  `model_path` may point to a safetensors
    or torch checkpoint with open_clip names; without it the server refuses to start unless
    `synthetic_weights` is set (random weights: only useful for contract / throughput tests)."""
    from .siglip import SiglipImageEngine, synthetic_state_dict, SO400M_384
    cfg = dict(SO400M_384)
    cfg.update(config.get("model_config", {}))
    state = state if state is not None else _load_state(config)
    if state is None:
        if not config.get("synthetic_weights"):
            raise SystemExit("config needs model_path (open_clip checkpoint) or synthetic_weights: true")
        state = synthetic_state_dict(cfg, seed=int(config.get("synthetic_weights_seed", 0x5EED0005)))
    eng = SiglipImageEngine.from_state_dict(state, cfg, max_batch=int(config["max_batch_size"]),
                                            gelu=config.get("gelu", "erf"), eps=float(config.get("layer_norm_eps", 1e-6)))
    eng.image_size = (cfg["img_size"], cfg["img_size"])
    return eng


def load_text_engine(config, state=None):
    """The text tower (`model.encode_text`, clip_server.py:98) and its tokenizer.  Returns (engine, tokenizer);
    the tokenizer is None when the config has no `tokenizer_path` (text requests then answer 500)."""
    from .siglip import SiglipTextEngine, synthetic_text_state_dict, SO400M_TEXT
    cfg = dict(SO400M_TEXT)
    cfg.update(config.get("text_config", {}))
    state = state if state is not None else _load_state(config)
    if state is None:
        if not config.get("synthetic_weights"):
            raise SystemExit("config needs model_path (open_clip checkpoint) or synthetic_weights: true")
        state = synthetic_text_state_dict(cfg, seed=int(config.get("synthetic_weights_seed", 0x5EED0005)) + 1)
    eng = SiglipTextEngine.from_state_dict(state, cfg, max_batch=int(config["max_batch_size"]),
                                           gelu=config.get("gelu", "erf"), eps=float(config.get("layer_norm_eps", 1e-6)))
    tok = None
    if config.get("tokenizer_path"):
        tok = SiglipTokenizer(config["tokenizer_path"], cfg["context_length"])
    return eng, tok


def main(argv):
    from aiohttp import web
    with open(argv[1], "r") as f:
        config = json.load(f)
    state = _load_state(config)
    text_engine, tokenizer = load_text_engine(config, state)
    server = ClipServer(config, load_engine(config, state), text_engine, tokenizer)
    print("Model loaded")
    server.start_threads()
    print("Ready")
    web.run_app(server.make_app(), host="", port=int(config["port"]), print=None)


if __name__ == "__main__":
    main(sys.argv)
