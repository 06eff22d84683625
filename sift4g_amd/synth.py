"""Synthetic protein sets for the BASELINE.json configs.

Deterministic, counter-based splitmix64; residues are drawn i.i.d. from the
reference's own background table aa_frequency[26]
(/root/reference/sift4g/src/constants.hpp:52-54; zeros for B, J, O, U, X, Z).
Everything is generated once here (numpy) and the SAME arrays are handed to the
HIP path and to the CPU oracle.
"""
import numpy as np

SEED_BASE = 0x51F74600
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# aa_frequency[26], letter-indexed A..Z (constants.hpp:52-54)
AA_FREQUENCY = np.array([
    0.07587, 0, 0.01666, 0.05287, 0.06377, 0.04096, 0.06847, 0.02246, 0.05817, 0, 0.05957, 0.09427, 0.02376,
    0.04456, 0, 0.04906, 0.03976, 0.05166, 0.07127, 0.05677, 0, 0.06587, 0.01236, 0, 0.03186, 0])
_CUM = np.cumsum(AA_FREQUENCY / AA_FREQUENCY.sum())
_CUM[-1] = 1.0
_THRESH = np.minimum(np.floor(_CUM * 4294967296.0), 4294967295.0).astype(np.uint64)  # on the top 32 bits
_LETTERS = np.nonzero(AA_FREQUENCY)[0].astype(np.uint8)


def splitmix64(seed, n, offset=0):
    """n outputs of the splitmix64 stream seeded with `seed`, starting at position `offset`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(offset + 1, offset + n + 1, dtype=np.uint64) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _uniform(seed, n, offset=0):
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def residues(seed, n, offset=0):
    """n residue codes drawn from AA_FREQUENCY."""
    top = splitmix64(seed, n, offset) >> np.uint64(32)
    return np.searchsorted(_THRESH, top, side="right").astype(np.uint8).clip(0, 25)


def query_lengths(seed, n, lo=300, hi=400):
    return (lo + splitmix64(seed, n) % np.uint64(hi - lo + 1)).astype(np.int64)


def target_lengths(seed, n, mu=5.68, sigma=0.55, lo=30, hi=3000):
    """lognormal(mu, sigma) clipped to [lo, hi] (mean ~340: Swiss-Prot-like)."""
    u1 = np.maximum(_uniform(seed, n), 1e-300)
    u2 = _uniform(seed ^ 0xA5A5A5A5, n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return np.clip(np.floor(np.exp(mu + sigma * z)), lo, hi).astype(np.int64)


def random_set(seed, lengths):
    """-> (codes uint8[total], offsets uint64[n+1])"""
    lengths = np.asarray(lengths, dtype=np.int64)
    off = np.zeros(len(lengths) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lengths).astype(np.uint64)
    total = int(off[-1])
    codes = np.empty(max(total, 1), dtype=np.uint8)
    step = 1 << 22

    def fill(b):
        m = min(step, total - b)
        codes[b:b + m] = residues(seed, m, b)

    starts = range(0, total, step)
    if total > 4 * step:        # counter-based generator: chunks are independent, numpy releases the GIL inside its loops
        import os
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(fill, starts))
    else:
        for b in starts:
            fill(b)
    return codes, off


def mutate_batch(seed, src, identity, indel):
    """len(identity) homologs of `src`: each position substituted with prob 1-identity[h], deleted /
    followed by an insertion with prob indel[h]/2 each.  -> (codes of all homologs concatenated, lengths)"""
    n, m = len(src), len(identity)
    u = _uniform(seed, 3 * n * m).reshape(3, m, n)
    sub = residues(seed ^ 0x5151, 2 * n * m).reshape(2, m, n)
    ident = np.asarray(identity, dtype=np.float64)[:, None]
    half = np.asarray(indel, dtype=np.float64)[:, None] * 0.5
    out = np.where(u[0] < ident, src[None, :], sub[0])
    keep = u[1] >= half
    ins = u[2] < half
    # row-major interleave: residue i (if kept) followed by an inserted residue (if ins[i])
    cnt = keep.astype(np.int64) + ins.astype(np.int64)
    flat = cnt.ravel()
    pos = (np.cumsum(flat) - flat).reshape(m, n)
    res = np.empty(int(flat.sum()), dtype=np.uint8)
    res[pos[keep]] = out[keep]
    res[(pos + keep)[ins]] = sub[1][ins]
    return res, cnt.sum(axis=1)


# The length tail of a real Swiss-Prot that SURVEY.md 8d's generator clips away at 3000 (database_alignment.cpp:36-47 streams
# whatever the FASTA holds): the body's lognormal runs on unclipped, a thin second component fills the 2 000 .. 10 000 range
# (0.3 % of the sequences, lognormal(7.6, 0.5): median 2 000), and a handful of giants are planted: just below and just above
# the 12 288 residues beyond which s4g_db_create gives sequences a tile of their own, up to titin's 35 213.
SWISSPROT_TAIL = dict(hi=36000, frac=0.003, mu=7.6, sigma=0.5,
                      giants=(10200, 12288, 12289, 14507, 18141, 22152, 27118, 35213))


def make_config(cfg=2, n_queries=1000, n_targets=100000, homologs_per_query=50, q_len=(300, 400),
                t_len=dict(mu=5.68, sigma=0.55, lo=30, hi=3000), tail=None):
    """Synthetic workload of BASELINE.json configs 2/3/5.

    cfg 2: 1k queries U[300,400] x 100k targets; cfg 3: same generator at 570k targets;
    cfg 5: long queries (pass q_len=(2000, 5000), n_queries=64).
    tail: a dict like SWISSPROT_TAIL -- the same set with a realistic length tail instead of the clip at 3000 (the sequences
    that are not in the tail keep the lengths and residues' seeds of the clipped set); `giants` lists the planted ids.
    Returns dict(q_codes, q_off, t_codes, t_off, planted) with planted = list of (query, target) ids.
    """
    seed = SEED_BASE + cfg
    ql = query_lengths(seed + 0x100, n_queries, *q_len)
    q_codes, q_off = random_set(seed + 0x200, ql)
    giants = []
    if tail:
        t_len = dict(t_len, hi=tail["hi"])
    tl = target_lengths(seed + 0x300, n_targets, **t_len)
    if tail:
        pick = _uniform(seed + 0x700, n_targets) < tail["frac"]
        heavy = target_lengths(seed + 0x800, n_targets, mu=tail["mu"], sigma=tail["sigma"], lo=t_len["lo"], hi=tail["hi"])
        tl = np.where(pick, heavy, tl)
    planted = []
    n_h = min(homologs_per_query, n_targets // max(n_queries, 1)) if n_queries else 0
    hom = {}
    if n_h > 0:
        stride = max(1, n_targets // (n_queries * n_h))
        ident = 0.30 + 0.60 * _uniform(seed + 0x400, n_queries * n_h)
        indel = 0.05 * _uniform(seed + 0x500, n_queries * n_h)
        for q in range(n_queries):
            src = q_codes[int(q_off[q]):int(q_off[q + 1])]
            k0 = q * n_h
            seqs, lens = mutate_batch(seed + 0x10000 + q, src, ident[k0:k0 + n_h], indel[k0:k0 + n_h])
            ends = np.cumsum(lens)
            for h in range(n_h):
                t = (k0 + h) * stride
                hom[t] = seqs[ends[h] - lens[h]:ends[h]]
                tl[t] = lens[h]
                planted.append((q, t))
    if tail:
        # the giants sit at fixed, spread-out ids, moved on past any id a planted homolog uses
        step = max(1, n_targets // (len(tail["giants"]) + 1))
        for k, length in enumerate(tail["giants"]):
            t = min(n_targets - 1, (k + 1) * step)
            while t in hom or t in giants:
                t += 1
            if t < n_targets:
                tl[t] = length
                giants.append(int(t))
    t_codes, t_off = random_set(seed + 0x600, tl)
    for t, seq in hom.items():
        t_codes[int(t_off[t]):int(t_off[t + 1])] = seq
    # a giant that is all noise would never be looked at: each carries a copy of one query somewhere inside it
    for k, t in enumerate(giants):
        if n_queries == 0:
            break
        q = (k * 37) % n_queries
        src = q_codes[int(q_off[q]):int(q_off[q + 1])]
        at = int(t_off[t]) + int(tl[t]) * (k + 1) // (len(giants) + 2)
        if at + len(src) <= int(t_off[t + 1]):
            t_codes[at:at + len(src)] = src
            planted.append((q, t))
    return dict(q_codes=q_codes, q_off=q_off, t_codes=t_codes, t_off=t_off, planted=planted, giants=giants)


def device_database(n_targets, seed, device, mu=5.68, sigma=0.55, lo=30, hi=3000, chunk=1 << 28):
    """BASELINE configs[3]: a database shard generated ON THE DEVICE (150M sequences never visit the host).
    Same length law and residue frequencies as make_config, torch's device generator seeded with `seed`.
    -> (codes uint8 [total] , offsets int64 [n+1]) torch tensors on `device`."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
    z = torch.randn(n_targets, device=device, generator=g)
    lengths = torch.exp(z * sigma + mu).floor_().clamp_(lo, hi).to(torch.int64)
    offsets = torch.zeros(n_targets + 1, dtype=torch.int64, device=device)
    torch.cumsum(lengths, 0, out=offsets[1:])
    total = int(offsets[-1])
    cum = torch.from_numpy(_CUM[:-1].astype(np.float32)).to(device)       # letter = number of cumulative bounds <= u
    codes = torch.empty(total, dtype=torch.uint8, device=device)
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        u = torch.rand(b - a, device=device, generator=g)
        codes[a:b] = torch.bucketize(u, cum, right=True).clamp_(0, 25).to(torch.uint8)
    return codes, offsets
