"""Seeded synthetic texts and pattern sets for the BASELINE.json configurations.

Everything is generated locally (no corpus is available offline): numpy PCG64 streams with the
seed stated by the caller, so the GPU box and the build container produce identical bytes.

Alphabet convention of the reference: a pattern/text symbol is an `alpha_t` (uint16) equal to
byte + 5 (CHARACTER_OFFSET, /root/reference/src/main/index_types.h:61-66).
"""
import numpy as np

CHARACTER_OFFSET = 5
ALPHA_SIZE = 261


def t_acgt(n, seed):
    """T_acgt(N, seed): N bytes uniform over {A,C,G,T}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 26
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s:s + m] = lut[rng.integers(0, 4, m, dtype=np.uint8)]
    return out


def _vocabulary(rng, nwords):
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2,
                  2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])
    p = p / p.sum()
    lens = np.clip(rng.poisson(4.2, nwords) + 1, 1, 14)
    words = []
    for i in range(nwords):
        words.append(letters[rng.choice(26, int(lens[i]), p=p)].tobytes())
    return words


def _eng_tables(rng, nwords):
    words = _vocabulary(rng, nwords)
    seps = [b" "] * 70 + [b", "] * 8 + [b". "] * 7 + [b".\n"] * 3 + [b"; "] + [b": "] + [b"? "] + [b"! "] + \
           [b" - "] + [b" (", b") ", b" \"", b"\" ", b"'s "] + [b"\n\n"] + [b"/", b"_", b"=", b"+", b"*", b"&", b"%",
                                                                           b"$", b"#", b"@", b"~", b"^", b"|", b"\\",
                                                                           b"<", b">", b"[", b"]", b"{", b"}", b"`"]
    # token pool: plain words, Capitalised words, UPPER words, numbers
    pool_tokens = []
    for w in words:
        pool_tokens.append(w)
    ncap = nwords // 8
    for w in words[:ncap]:
        pool_tokens.append(w[:1].upper() + w[1:])
    for w in words[:ncap // 8]:
        pool_tokens.append(w.upper())
    for i in range(512):
        pool_tokens.append(str(int(rng.integers(0, 10 ** int(rng.integers(1, 6))))).encode())
    ntok = len(pool_tokens)
    # Zipf-like weights over plain words; fixed small mass for the other classes
    w_plain = 1.0 / np.arange(1, nwords + 1) ** 1.05
    w_cap = 0.08 * w_plain[:ncap]
    w_up = 0.01 * w_plain[:ncap // 8]
    w_num = np.full(512, 0.02 * w_plain.sum() / 512)
    weights = np.concatenate([w_plain, w_cap, w_up, w_num])
    weights /= weights.sum()
    cdf = np.cumsum(weights)
    tok_len = np.array([len(t) for t in pool_tokens], dtype=np.int64)
    tok_start = np.zeros(ntok, dtype=np.int64)
    tok_start[1:] = np.cumsum(tok_len[:-1])
    tok_pool = np.frombuffer(b"".join(pool_tokens), dtype=np.uint8)
    sep_len = np.array([len(s) for s in seps], dtype=np.int64)
    sep_start = np.zeros(len(seps), dtype=np.int64)
    sep_start[1:] = np.cumsum(sep_len[:-1])
    sep_pool = np.frombuffer(b"".join(seps), dtype=np.uint8)
    return dict(cdf=cdf, ntok=ntok, tok_len=tok_len, tok_start=tok_start, tok_pool=tok_pool,
                sep_len=sep_len, sep_start=sep_start, sep_pool=sep_pool, nseps=len(seps))


def t_eng(n, seed, nwords=20000):
    """T_eng(N, seed): deterministic English-like text, sigma ~ 96 printable ASCII + newline:
    Zipfian synthetic vocabulary, sentence capitalisation, digits, punctuation, newlines."""
    rng = np.random.Generator(np.random.PCG64(seed))
    T = _eng_tables(rng, nwords)
    cdf, ntok, tok_len, tok_start, tok_pool = T["cdf"], T["ntok"], T["tok_len"], T["tok_start"], T["tok_pool"]
    sep_len, sep_start, sep_pool = T["sep_len"], T["sep_start"], T["sep_pool"]
    seps = range(T["nseps"])

    out = np.empty(n, dtype=np.uint8)
    pos = 0
    chunk_tokens = 1 << 20
    while pos < n:
        ids = np.searchsorted(cdf, rng.random(chunk_tokens), side="right").clip(0, ntok - 1)
        sids = rng.integers(0, len(seps), chunk_tokens)
        lens = np.empty(2 * chunk_tokens, dtype=np.int64)
        starts = np.empty(2 * chunk_tokens, dtype=np.int64)
        lens[0::2] = tok_len[ids]
        lens[1::2] = sep_len[sids]
        starts[0::2] = tok_start[ids]
        starts[1::2] = sep_start[sids] + len(tok_pool)
        pool = np.concatenate([tok_pool, sep_pool])
        total = int(lens.sum())
        dst0 = np.zeros(2 * chunk_tokens, dtype=np.int64)
        dst0[1:] = np.cumsum(lens[:-1])
        src = np.repeat(starts - dst0, lens) + np.arange(total, dtype=np.int64)
        piece = pool[src]
        m = min(total, n - pos)
        out[pos:pos + m] = piece[:m]
        pos += m
    return out


def t_eng_torch(n, seed, device, nwords=20000):
    """Same token tables and distribution as t_eng, sampled with torch on `device` (seconds instead of
    minutes for 1 GiB).  Deterministic for a given (seed, device type); NOT byte-identical to t_eng."""
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    T = _eng_tables(rng, nwords)
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cdf = torch.from_numpy(T["cdf"]).to(dev)
    tok_len = torch.from_numpy(T["tok_len"]).to(dev)
    tok_start = torch.from_numpy(T["tok_start"]).to(dev)
    npool = len(T["tok_pool"])
    sep_len = torch.from_numpy(T["sep_len"]).to(dev)
    sep_start = torch.from_numpy(T["sep_start"]).to(dev) + npool
    pool = torch.from_numpy(np.concatenate([T["tok_pool"], T["sep_pool"]])).to(dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    pos = 0
    chunk = 1 << 22
    while pos < n:
        ids = torch.searchsorted(cdf, torch.rand(chunk, generator=g, device=dev, dtype=torch.float64), right=True).clamp_(0, T["ntok"] - 1)
        sids = torch.randint(0, T["nseps"], (chunk,), generator=g, device=dev)
        lens = torch.stack([tok_len[ids], sep_len[sids]], dim=1).reshape(-1)
        starts = torch.stack([tok_start[ids], sep_start[sids]], dim=1).reshape(-1)
        ends = torch.cumsum(lens, 0)
        total = int(ends[-1].item())
        dst0 = ends - lens
        src = torch.repeat_interleave(starts - dst0, lens) + torch.arange(total, device=dev)
        m = min(total, n - pos)
        out[pos:pos + m] = pool[src[:m]]
        pos += m
    return out.cpu().numpy()


def t_counter(n):
    """The reference tests' 'interesting to search for' text (generate_text,
    /root/reference/src/main/index_test_funcs.c:316-338), restated: base-6 counter digits a..f."""
    out = np.full(n, ord("x"), dtype=np.uint8)
    i = 0
    j = 0
    while i < n:
        k = j
        while k > 0:
            out[i] = ord("a") + (k - 1) % 6
            i += 1
            if i == n:
                return out
            k //= 6
        j += 1
    return out


def to_alpha(b):
    return np.asarray(b, dtype=np.uint8).astype(np.uint16) + CHARACTER_OFFSET


def p_rand(k, n, seed, alphabet=b"ACGT"):
    """P_rand(k, n, seed): n patterns of length k uniform over `alphabet`; returns (plen, flat)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lut = np.frombuffer(bytes(alphabet), dtype=np.uint8).astype(np.uint16) + CHARACTER_OFFSET
    flat = lut[rng.integers(0, len(lut), n * k, dtype=np.uint8 if len(lut) <= 256 else np.int64)]
    return np.full(n, k, dtype=np.int32), np.ascontiguousarray(flat, dtype=np.uint16)


def p_hit(kmin, kmax, n, seed, text):
    """P_hit: n substrings of `text` sampled at uniform offsets, lengths uniform in [kmin,kmax]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    plen = rng.integers(kmin, kmax + 1, n).astype(np.int32)
    start = rng.integers(0, len(text) - kmax, n).astype(np.int64)
    tot = int(plen.sum())
    dst0 = np.zeros(n, dtype=np.int64)
    dst0[1:] = np.cumsum(plen[:-1])
    src = np.repeat(start - dst0, plen) + np.arange(tot, dtype=np.int64)
    flat = text[src].astype(np.uint16) + CHARACTER_OFFSET
    return plen, np.ascontiguousarray(flat)


def starts_of(plen):
    s = np.zeros(len(plen), dtype=np.int64)
    if len(plen):
        s[1:] = np.cumsum(plen[:-1], dtype=np.int64)
    return s
