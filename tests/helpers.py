"""Seeded synthetic inputs shared by tests, smoke and bench."""
import math

import numpy as np
import torch


def bitrev(v, nbits):
    r = np.zeros_like(v)
    for b in range(nbits):
        r |= ((v >> b) & 1) << (nbits - 1 - b)
    return r


def fps_closed_form(p, m):
    """argmax of the running min-distance, ties -> min (bitrev(k mod bs), k)."""
    p = np.asarray(p, np.float32)
    n = len(p)
    L = min(9, int(math.floor(math.log2(n))))
    bs = 1 << L
    k = np.arange(n, dtype=np.int64)
    key = bitrev(k % bs, L).astype(np.int64) * (1 << 32) + k
    tmp = np.full(n, 1e10, np.float32)
    out = [0]
    old = 0
    for _ in range(1, m):
        d = p - p[old]
        d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        tmp = np.minimum(d, tmp)
        c = np.where(tmp == tmp.max())[0]
        old = int(c[np.argmin(key[c])])
        out.append(old)
    return np.array(out, np.int32)


def object_cloud(gen, n, with_replacement=False):
    """Visible half of an ellipsoid surface + 1 mm noise, camera frame, metres."""
    axes = 0.04 + 0.08 * torch.rand(3, generator=gen)
    v = torch.randn(n, 3, generator=gen)
    v = v / v.norm(dim=1, keepdim=True)
    v[:, 2] = -v[:, 2].abs()
    p = v * axes
    t = torch.tensor([0.0, 0.0, 0.85]) + (torch.rand(3, generator=gen) - 0.5) * torch.tensor([0.2, 0.2, 0.7])
    p = p + t + 1e-3 * torch.randn(n, 3, generator=gen)
    if with_replacement:
        sel = torch.randint(0, max(8, n // 5), (n,), generator=gen)
        p = p[sel]
    return p.float().contiguous()


def constructed_similarity(perm, N, gen, n_bg=40, hi=8.0, noise=1.5):
    """A (B,N+1,N+1) similarity (cosine / temp scale) whose row 1+i matches column
    1+perm[b,i]; the last `n_bg` rows (and their partner columns) match the
    background token 0 instead.  Returns (atten, score (B,2N) in (0,1)).
    Needed because random weights give all-background assignments."""
    B = perm.shape[0]
    atten = noise * (torch.rand(B, N + 1, N + 1, generator=gen) * 2 - 1)
    score = torch.empty(B, 2 * N)
    for b in range(B):
        s1 = 0.8 + 0.19 * torch.rand(N, generator=gen)
        s2 = 0.8 + 0.19 * torch.rand(N, generator=gen)
        for i in range(N):
            j = int(perm[b, i])
            if i < N - n_bg:
                atten[b, 1 + i, 1 + j] = hi + torch.rand((), generator=gen)
            else:
                atten[b, 1 + i, 0] = hi + torch.rand((), generator=gen)
                atten[b, 0, 1 + j] = hi + torch.rand((), generator=gen)
                s1[i] = 0.05 + 0.1 * torch.rand((), generator=gen)
                s2[j] = 0.05 + 0.1 * torch.rand((), generator=gen)
        score[b, :N] = s1
        score[b, N:] = s2
    return atten, score


def seeded_checked(shape, seed, checksum, scale=1.0):
    """Regenerate a fixture input that was stored as a seed (tests/golden/make_golden.py::seeded) and verify
    it against the checksum recorded when the reference ran on it."""
    t = scale * torch.randn(*shape, generator=torch.Generator().manual_seed(int(seed)))
    d = t.double().flatten()
    got = np.array([d.sum().item(), (d * torch.arange(1, d.numel() + 1, dtype=torch.float64)).sum().item() / d.numel()])
    assert np.allclose(got, np.asarray(checksum), rtol=1e-9, atol=1e-9), "seeded fixture input differs from the one the reference saw"
    return t


def tensor_checksum(t):
    d = t.double().flatten()
    return np.array([d.sum().item(), (d * torch.arange(1, d.numel() + 1, dtype=torch.float64)).sum().item() / d.numel()])


from unopose_amd.synthetic import congruent_pair, random_rotation  # noqa: E402,F401
