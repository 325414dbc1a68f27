"""Host mirror of the wire helpers in src/common.rs that define the boundary of the scoring path."""
import numpy as np


def decode_fp16_buffer(buf: bytes) -> np.ndarray:
    """common.rs:98-102: little-endian fp16 bytes -> f32 vector."""
    return np.frombuffer(buf, dtype="<f2").astype(np.float32)


def chunk_fp16_buffer(buf: bytes) -> np.ndarray:
    """common.rs:104-108: little-endian fp16 bytes -> f16 bit patterns."""
    return np.frombuffer(buf, dtype="<u2").copy()


def get_total_embedding(terms, embedding_size, query_server, predefined_embeddings=None):
    """common.rs:215-274.  `terms` are dicts with optional keys image (bytes, already resized),
    text, embedding (list of f32), predefined_embedding (name) and weight.  `query_server` is
    called with {"images": [...]} and/or {"text": [...]} and returns a list of fp16 byte strings
    (the clip_server contract).  The sum is NOT re-normalised."""
    total = np.zeros(embedding_size, np.float32)
    images, image_w, texts, text_w = [], [], [], []
    predefined_embeddings = predefined_embeddings or {}
    for term in terms:
        w = np.float32(term.get("weight", 1.0) if term.get("weight") is not None else 1.0)
        if term.get("image") is not None:
            images.append(term["image"])
            image_w.append(w)
        if term.get("text") is not None:
            texts.append(term["text"])
            text_w.append(w)
        if term.get("embedding") is not None:
            total += np.asarray(term["embedding"], np.float32) * w
        name = term.get("predefined_embedding")
        if name is not None and name in predefined_embeddings:
            total = total + np.asarray(predefined_embeddings[name], np.float32) * w
    batches = []
    if images:
        batches.append(({"images": images}, image_w))
    if texts:
        batches.append(({"text": texts}, text_w))
    for batch, weights in batches:
        embs = query_server(batch)
        for emb, w in zip(embs, weights):
            total += decode_fp16_buffer(emb) * w
    return total
