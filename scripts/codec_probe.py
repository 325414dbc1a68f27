"""Developer probe (needs a GPU): what kind of 64 x 8-bit codec the HARD synthetic set needs.  PQ-only recall@10 (top-10 by the ADC
estimate against the exact top-10) and recall@10 inside the ADC top-100 / top-400, for the codec variants below, each assigned by the
reference's rule (the centroid with the largest dot product, diskann/src/vector.rs quantize_batch).  Pure torch (no library calls).

    python scripts/codec_probe.py [rows] [kind]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "meme-search-engine_amd")]
import torch  # noqa: E402
import bench_ann as ba  # noqa: E402

D, M, DPC, KC = 1152, 64, 18, 256


def rotation_random(x, seed=4):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.linalg.qr(torch.randn(D, D, device="cuda", generator=g))[0]          # rows = new axes


def rotation_pca_balanced(x):
    """eigenvectors of the second-moment matrix, dealt to the 64 subspaces so that the products of their eigenvalues are level (the
    eigenvalue allocation of OPQ's parametric solution)"""
    m2 = (x.T @ x) / x.shape[0]
    lam, vec = torch.linalg.eigh(m2.double())
    order = torch.argsort(lam, descending=True)
    lam, vec = lam[order].clamp_min(1e-12), vec[:, order]
    logs = [0.0] * M
    fill = [[] for _ in range(M)]
    ll = (torch.log(lam) - torch.log(lam[-1])).tolist()      # >= 0: a subspace's level = how much variance (in log terms) it holds so far
    for d in range(D):
        b = min((b_ for b_ in range(M) if len(fill[b_]) < DPC), key=lambda b_: logs[b_])
        fill[b].append(d)
        logs[b] += ll[d]
    perm = [d for b in range(M) for d in fill[b]]
    return vec[:, perm].T.float().contiguous()


def kmeans(ts, rule, iters, seed=5):
    """per-subspace centroids [M, KC, DPC]; rule: 'mip' (assign by largest dot, update = mean), 'l2', 'sph' (spherical: unit
    directions, assignment by largest dot, every centroid scaled to the mean projection of its members)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    S = ts.shape[0]
    sub = ts.view(S, M, DPC).permute(1, 0, 2).contiguous()                               # [M, S, DPC]
    c = sub[:, torch.randperm(S, device="cuda", generator=g)[:KC]].clone()              # [M, KC, DPC]
    for _ in range(iters):
        if rule == "sph":
            cu = c / c.norm(dim=2, keepdim=True).clamp_min(1e-12)
            dots = torch.bmm(sub, cu.transpose(1, 2))
            a = dots.argmax(dim=2)
        elif rule == "l2":
            dots = torch.bmm(sub, c.transpose(1, 2)) - 0.5 * (c * c).sum(dim=2)[:, None, :]
            a = dots.argmax(dim=2)
        else:
            a = torch.bmm(sub, c.transpose(1, 2)).argmax(dim=2)
        onehot_sum = torch.zeros(M, KC, DPC, device="cuda")
        onehot_sum.scatter_add_(1, a[:, :, None].expand(-1, -1, DPC), sub)
        cnt = torch.zeros(M, KC, device="cuda").scatter_add_(1, a, torch.ones_like(a, dtype=torch.float32))
        mean = onehot_sum / cnt.clamp_min(1)[:, :, None]
        if rule == "sph":
            u = mean / mean.norm(dim=2, keepdim=True).clamp_min(1e-12)
            proj = torch.zeros(M, KC, device="cuda").scatter_add_(1, a, (sub * u.gather(1, a[:, :, None].expand(-1, -1, DPC))).sum(dim=2))
            mean = u * (proj / cnt.clamp_min(1))[:, :, None]
        c = torch.where((cnt > 0)[:, :, None], mean, c)
    return c


def encode(rows_f16, T, c):
    """codes [n, M] by the reference's rule: the centroid with the largest dot product"""
    n = rows_f16.shape[0]
    codes = torch.empty(n, M, dtype=torch.uint8, device="cuda")
    for i in range(0, n, 1 << 16):
        t = (rows_f16[i:i + (1 << 16)].float() @ T.T).view(-1, M, DPC).permute(1, 0, 2)
        codes[i:i + (1 << 16)] = torch.bmm(t, c.transpose(1, 2)).argmax(dim=2).T.to(torch.uint8)
    return codes


def adc_topk(q_f16, T, c, codes, k):
    lut = torch.bmm((q_f16.float() @ T.T).view(-1, M, DPC).permute(1, 0, 2), c.transpose(1, 2))   # [M, nq, KC]
    nq, n = q_f16.shape[0], codes.shape[0]
    best_s = torch.full((nq, k), -1e30, device="cuda")
    best_i = torch.zeros(nq, k, dtype=torch.long, device="cuda")
    for i in range(0, n, 1 << 17):
        cc = codes[i:i + (1 << 17)].long()                                              # [b, M]
        s = torch.zeros(nq, cc.shape[0], device="cuda")
        for m in range(M):
            s += lut[m][:, cc[:, m]]
        ts, ti = torch.topk(torch.cat([best_s, s], dim=1), k, dim=1)
        src = torch.cat([best_i, torch.arange(i, i + cc.shape[0], device="cuda")[None, :].expand(nq, -1)], dim=1)
        best_s, best_i = ts, src.gather(1, ti)
    return best_i


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
    kind = sys.argv[2] if len(sys.argv) > 2 else "hard"
    if kind == "easy":
        gen = ba.easy_generator(n)
        rows, q = gen(n, 1), gen(256, 2)
    else:
        hs = ba.HardSet(n, **ba.HARD_PARAMS)
        rows = hs.rows(n, 1)
        q = hs.rows(256, 2) if kind == "hard" else hs.rows(256, 2, queries="ood", gap=ba.OOD_GAP, extra_noise=ba.OOD_EXTRA_NOISE)
    exact = torch.topk(q.float() @ rows.float().T, 10, dim=1).indices
    g = torch.Generator(device="cuda").manual_seed(9)

    def sample(s):
        return rows[torch.randperm(n, device="cuda", generator=g)[:s]].float()

    def report(name, T, c, t_train):
        codes = encode(rows, T, c)
        top = adc_topk(q, T, c, codes, 400)
        hit = (top[:, :, None] == exact[:, None, :])                                   # [nq, 400, 10]
        r10 = hit[:, :10].any(dim=1).float().mean().item()
        r100 = hit[:, :100].any(dim=1).float().mean().item()
        r400 = hit.any(dim=1).float().mean().item()
        used = torch.stack([torch.bincount(codes[:, m].long(), minlength=KC).gt(0).sum() for m in range(0, M, 8)]).float().mean().item()
        print(f"{name:58s} ADC-only recall@10 {r10:.3f}   exact top-10 inside ADC top-100 {r100:.3f}   top-400 {r400:.3f}   "
              f"centroids in use {used:.0f}/256   trained in {t_train:.1f} s", flush=True)

    print(f"# {kind} set, {n} rows, 256 queries", flush=True)
    for name, rot, ssz, rule, iters in (("random rotation, 20 k sample, mip k-means x3 (the bench's codec)", "rand", 20_000, "mip", 3),
                                        ("random rotation, 200 k sample, mip k-means x10", "rand", 200_000, "mip", 10),
                                        ("balanced PCA rotation, 200 k sample, mip k-means x10", "pca", 200_000, "mip", 10),
                                        ("balanced PCA rotation, 200 k sample, spherical k-means x10", "pca", 200_000, "sph", 10),
                                        ("balanced PCA rotation, 200 k sample, l2 k-means x10 (mip-assigned)", "pca", 200_000, "l2", 10)):
        t0 = time.perf_counter()
        x = sample(ssz)
        T = rotation_random(x) if rot == "rand" else rotation_pca_balanced(x)
        c = kmeans(x @ T.T, rule, iters)
        torch.cuda.synchronize()
        report(name, T, c, time.perf_counter() - t0)


if __name__ == "__main__":
    main()
