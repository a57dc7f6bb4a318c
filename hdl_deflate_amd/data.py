"""Synthetic inputs shaped like the reference's test data families (test_deflate.py:38-66).

family 1  "   Hello World! <i>     " joined by " "      (text + running counter)
family 2  "Hi: <rand 0..4095> " joined by " "
family 3  uniform random bytes
family 4  random '0'/'1' characters
`family_bytes` is the exact small-scale Python form (seeded); `make_blocks` builds millions of
distinct blocks directly on the GPU with the same textual shape (SURVEY 8(d) cfg 2: block b is
family 1 + (b mod 4); family 1's counter starts at b*16 so blocks differ; the working set must be
far larger than the 256 MB Infinity Cache -- no tiling of a small pool)."""
import random


def family_bytes(f, n, seed=1, counter0=0):
    r = random.Random(seed)
    if f == 1:
        s = " ".join("   Hello World! " + str(counter0 + i) + "     " for i in range(n // 10 + 2)).encode()
    elif f == 2:
        s = " ".join("Hi: " + str(r.randrange(0, 0x1000)) + " " for _ in range(n // 4 + 2)).encode()
    elif f == 3:
        s = bytes(r.randrange(256) for _ in range(n))
    elif f == 4:
        s = "".join(str(r.randrange(2)) for _ in range(n)).encode()
    else:
        raise ValueError("family %r" % (f,))
    return s[:n]


def _ndigits(torch, v):
    return 1 + (v >= 10).long() + (v >= 100).long() + (v >= 1000).long() + (v >= 10000).long() + \
        (v >= 100000).long() + (v >= 1000000).long() + (v >= 10000000).long() + (v >= 100000000).long()


def _item_text(torch, numbers, prefix, suffix_len, n, device):
    """rows of `prefix + str(number) + ' '*suffix_len` items, cut to n bytes per row -> uint8 [B, n]"""
    B, K = numbers.shape
    L = _ndigits(torch, numbers)
    plen = len(prefix)
    ilen = plen + L + suffix_len
    ends = torch.cumsum(ilen, dim=1)                                   # [B, K]
    pos = torch.arange(n, device=device).unsqueeze(0).expand(B, n).contiguous()
    idx = torch.searchsorted(ends, pos, right=True).clamp_(max=K - 1)  # item containing each position
    start = torch.gather(ends - ilen, 1, idx)
    o = pos - start                                                    # offset inside the item
    num = torch.gather(numbers, 1, idx)
    Ln = torch.gather(L, 1, idx)
    pfx = torch.tensor(list(prefix.encode()), dtype=torch.uint8, device=device)
    ch = torch.full((B, n), 32, dtype=torch.uint8, device=device)      # spaces
    in_p = o < plen
    ch[in_p] = pfx[o[in_p]]
    in_d = (o >= plen) & (o < plen + Ln)
    k = (plen + Ln - 1 - o).clamp_(min=0)                               # power of ten of this digit
    p10 = torch.tensor([10 ** e for e in range(10)], dtype=torch.int64, device=device)
    dig = (num // p10[k.clamp(max=9)]) % 10
    ch[in_d] = (dig[in_d] + 48).to(torch.uint8)
    return ch


def _mix32(torch, x):
    """murmur3's 32-bit finaliser on int64 tensors holding 32-bit values (a bijection: distinct counters -> distinct words)"""
    m = 0xFFFFFFFF
    x = (x ^ (x >> 16)) * 0x85EBCA6B & m
    x = (x ^ (x >> 13)) * 0xC2B2AE35 & m
    return x ^ (x >> 16)


def _block_words(torch, gsel, K, salt, device):
    """counter-based random words: int64 [rows, K] in [0, 2^32), a function of (seed salt, GLOBAL block index, k) only"""
    ctr = (gsel.unsqueeze(1) * 0x9E3779B1 + salt) & 0xFFFFFFFF
    k = torch.arange(K, device=device, dtype=torch.int64).unsqueeze(0)
    return _mix32(torch, _mix32(torch, ctr) + k * 0x632BE5AB & 0xFFFFFFFF)


def make_blocks(nblocks, n, device, seed=0, families=(1, 2, 3, 4), first_block=0, chunk=16384):
    """uint8 [nblocks, n] on `device`; block b (global index first_block + b) is family
    families[b % len(families)], every block distinct.  The content of a block is a function of (seed, global
    block index) alone (counter-based generator): any shard [b0, b1) of a job -- whatever the world size or the chunk
    grid -- holds exactly the bytes the single-GPU job holds at [b0, b1)."""
    import torch
    out = torch.empty((nblocks, n), dtype=torch.uint8, device=device)
    nf = len(families)
    # the text families build int64 [rows, n] intermediates: bound a chunk to 2^25 elements (16384 rows of 2 KiB -- the
    # default -- or 512 rows of 64 KiB).  The grid only bounds memory; it does not enter the contents.
    chunk = max(nf, min(chunk, (1 << 25) // max(n, 1)))
    for c0 in range(0, nblocks, chunk):
        c1 = min(nblocks, c0 + chunk)
        for fi, f in enumerate(families):
            gb = torch.arange(first_block + c0, first_block + c1, device=device)
            sel = torch.nonzero((gb % nf) == fi).squeeze(1)
            rows = c0 + sel
            gsel = gb[sel]
            B = rows.numel()
            if B == 0:
                continue
            salt = (seed * 1000003 + f * 7919 + 12345) & 0xFFFFFFFF
            if f == 3:
                w = _block_words(torch, gsel, (n + 3) // 4, salt, device)               # four bytes per word
                blk = w.unsqueeze(2) >> torch.tensor([0, 8, 16, 24], device=device)
                blk = (blk & 255).to(torch.uint8).reshape(B, -1)[:, :n]
            elif f == 4:
                w = _block_words(torch, gsel, (n + 31) // 32, salt, device)             # 32 characters per word
                blk = w.unsqueeze(2) >> torch.arange(32, device=device)
                blk = ((blk & 1) + 48).to(torch.uint8).reshape(B, -1)[:, :n]
            elif f == 2:
                K = n // 7 + 2
                nums = _block_words(torch, gsel, K, salt, device) >> 20                 # 0 .. 4095
                blk = _item_text(torch, nums, "Hi: ", 2, n, device)
            elif f == 1:
                K = n // 23 + 2
                nums = gsel.unsqueeze(1) * 16 + torch.arange(K, device=device).unsqueeze(0)
                blk = _item_text(torch, nums, "   Hello World! ", 6, n, device)
            else:
                raise ValueError("family %r" % (f,))
            out[rows] = blk
    return out


# ---- BASELINE configs[2]: enwik8 is not obtainable (no network), so an English-like substitute is generated:
# a fixed vocabulary with Zipf-distributed word frequencies, words separated by spaces, sentences by ". "
_SYLL = ("th", "e", "an", "in", "er", "on", "re", "at", "st", "en", "al", "or", "ti", "ar", "ng", "ou", "is", "it",
         "le", "ed", "ro", "ve", "co", "me", "de", "ha", "se", "li", "ra", "ne", "ic", "io", "ma", "ur", "wi", "ta")


def _vocab(size=4096, seed=12345):
    r = random.Random(seed)
    words, seen = [], set()
    while len(words) < size:
        w = "".join(r.choice(_SYLL) for _ in range(r.choice((1, 1, 2, 2, 2, 3, 3, 4))))
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


def make_text_blocks(nblocks, n, device, seed=0, vocab_size=4096):
    """uint8 [nblocks, n] of Zipf-distributed pseudo-English (enwik8 stand-in for BASELINE configs[2])."""
    import torch
    words = _vocab(vocab_size)
    maxw = max(len(w) for w in words) + 1
    tab = torch.zeros((vocab_size, maxw), dtype=torch.uint8)
    wl = torch.zeros(vocab_size, dtype=torch.int64)
    for k, w in enumerate(words):
        b = (w + " ").encode()
        tab[k, :len(b)] = torch.tensor(list(b), dtype=torch.uint8)
        wl[k] = len(b)
    tab, wl = tab.to(device), wl.to(device)
    ranks = torch.arange(1, vocab_size + 1, dtype=torch.float64, device=device)
    cdf = torch.cumsum(1.0 / ranks, 0)
    cdf = (cdf / cdf[-1]).to(torch.float32)
    g = torch.Generator(device=device)
    g.manual_seed(seed * 7919 + 17)
    out = torch.empty((nblocks, n), dtype=torch.uint8, device=device)
    K = n // 3 + 8                                       # words per block (average word+space > 3 bytes)
    chunk = max(1, (1 << 24) // n)
    pos = torch.arange(n, device=device).unsqueeze(0)
    for c0 in range(0, nblocks, chunk):
        c1 = min(nblocks, c0 + chunk)
        Bc = c1 - c0
        u = torch.rand((Bc, K), generator=g, device=device)
        wid = torch.searchsorted(cdf, u).clamp_(max=vocab_size - 1)
        ln = wl[wid]
        ends = torch.cumsum(ln, 1)
        p = pos.expand(Bc, n).contiguous()
        idx = torch.searchsorted(ends, p, right=True).clamp_(max=K - 1)
        o = p - torch.gather(ends - ln, 1, idx)
        w = torch.gather(wid, 1, idx)
        out[c0:c1] = tab[w, o.clamp_(max=maxw - 1)]
    return out
