/*
 * hdlz_oracle.c -- CPU restatement of the HDL-deflate hot path (plain C).
 *
 * TEST INFRASTRUCTURE.  This file is the checker, never the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (hdl_deflate_amd/csrc, libhdlz.so) has no CPU path and never links this file.
 *
 * Parity pin: this restatement is checked bit-for-bit against the JSON fixtures under tests/golden/, which were
 * produced by executing the UNMODIFIED reference /root/reference/deflate.py in the build
 * container (oracle/gen_golden.py; provenance: "reference source executed under a
 * clocked-only stand-in kernel, not under MyHDL 0.10"), and against stock zlib round trips.
 *
 * It follows the reference's state machine state by state (citations are file:line into
 * /root/reference/deflate.py) and is deliberately NOT shaped like the GPU kernels: the
 * match search walks distances, the distance code is found by the reference's linear walk,
 * Huffman codes come from a canonical-code builder -- so it is an independent check on the
 * closed-form arithmetic used on the device.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define HDLZ_OK 0
#define HDLZ_E_SHORT_INPUT 1   /* N < 5: the reference never starts (deflate.py:429-431,740-741) */
#define HDLZ_E_OUT_CAPACITY 2
#define HDLZ_E_BAD_BTYPE 3     /* deflate.py:719-721 */
#define HDLZ_E_BAD_DISTANCE 4  /* distance code 30/31, or distance reaching before the start */
#define HDLZ_E_NO_EOF 5        /* deflate.py:1535-1539 "NO EOF!" / input exhausted */
#define HDLZ_E_DYNAMIC_UNSUPPORTED 6
#define HDLZ_E_BAD_SYMBOL 7    /* literal/length symbols 286,287 (deflate.py:1437-1439 "< 1 bits") */
#define HDLZ_E_BAD_PARAM 8
#define HDLZ_E_HIP 9
#define HDLZ_E_BAD_TREE 10     /* dynamic block header describes an impossible Huffman code */

#define HDLZ_INFLATE_ASSUME_FIXED 1u /* DYNAMIC=False build: BTYPE ignored (deflate.py:724-732) */
#define HDLZ_INFLATE_ONEBLOCK 8u     /* ONEBLOCK=True build (deflate.py:40-41,:678,:1542,:1617) */

/* RFC1951 tables, as in deflate.py:100-110 (regenerated from the RFC, not pasted). */
static const uint16_t copy_length[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51,
                                         59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t extra_length_bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                              2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static uint16_t copy_distance[30];
static uint8_t extra_distance_bits[15];

static uint16_t out_codes[288];  /* bit-reversed fixed codes, deflate.py:112-149 */
static uint8_t code_length[288]; /* deflate.py:1066-1073 */
static uint16_t stat_leaves[512]; /* (sym<<4)|nbits, deflate.py:151-216 */
static int tables_ready = 0;

static unsigned rev_bits(unsigned b, int nb) { /* deflate.py:569-584 */
    unsigned r = 0;
    for (int i = 0; i < nb; i++) r |= ((b >> i) & 1u) << (nb - 1 - i);
    return r;
}

static void build_tables(void) {
    if (tables_ready) return;
    /* distance bases: two codes per extra-bit count */
    int base = 1;
    for (int c = 0; c < 30; c++) {
        int eb = c < 2 ? 0 : (c / 2 - 1);
        copy_distance[c] = (uint16_t)base;
        base += 1 << eb;
    }
    for (int i = 0; i < 15; i++) extra_distance_bits[i] = (uint8_t)(i == 0 ? 0 : i - 1);
    /* fixed literal/length code lengths, RFC1951 3.2.6 */
    for (int s = 0; s < 288; s++) code_length[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
    /* canonical codes */
    int bl_count[10] = {0}, next_code[10] = {0};
    for (int s = 0; s < 288; s++) bl_count[code_length[s]]++;
    int code = 0;
    for (int b = 1; b <= 9; b++) {
        code = (code + bl_count[b - 1]) << 1;
        next_code[b] = code;
    }
    for (int s = 0; s < 288; s++) {
        int l = code_length[s];
        unsigned c = (unsigned)next_code[l]++;
        out_codes[s] = (uint16_t)rev_bits(c, l);
    }
    /* 9-bit instant decode table; symbol 287's slot is 0 in the reference (deflate.py:212) */
    for (int s = 0; s < 288; s++) {
        int l = code_length[s];
        for (unsigned hi = 0; hi < (1u << (9 - l)); hi++)
            stat_leaves[out_codes[s] | (hi << l)] = (uint16_t)((s << 4) | l);
    }
    stat_leaves[483] = 0;
    tables_ready = 1;
}

size_t hdlz_oracle_out_bound(size_t n) { return 6 + (9 * n + 10 + 7) / 8; }

/* ------------------------------------------------------------------ compress (STARTC) */
typedef struct {
    uint8_t* out;
    size_t cap;
    size_t dout;   /* "do"  */
    unsigned doo;  /* "doo" */
    unsigned ob1;  /* carry byte */
    int overflow;
} bitw;

static void emit_byte(bitw* w, size_t at, unsigned v) {
    if (at < w->cap) w->out[at] = (uint8_t)v;
    else w->overflow = 1;
}

/* deflate.py:535-567: LSB-first append; a completed byte is written at "do". The reference
 * writes the partial byte on every put and completes it later; the final memory image is the same. */
static void put(bitw* w, unsigned d, unsigned width) {
    while (width) {
        unsigned take = 8 - w->doo;
        if (take > width) take = width;
        w->ob1 |= (d & ((1u << take) - 1)) << w->doo;
        d >>= take;
        width -= take;
        w->doo += take;
        if (w->doo == 8) { /* pshift / do_flush */
            emit_byte(w, w->dout, w->ob1);
            w->dout++;
            w->doo = 0;
            w->ob1 = 0;
        }
    }
}

typedef struct {
    uint32_t pos;
    uint16_t len;  /* 0 = literal */
    uint16_t dist; /* literal value when len==0 */
} hdlz_token;

static int compress_core(const uint8_t* x, size_t n, int cwindow, int maxmatch, uint8_t* out, size_t cap,
                         size_t* out_len, hdlz_token* toks, size_t tok_cap, size_t* ntok) {
    build_tables();
    if (cwindow < 1 || cwindow > 256 || (maxmatch != 5 && maxmatch != 10)) return HDLZ_E_BAD_PARAM;
    if (n < 5) return HDLZ_E_SHORT_INPUT; /* R0: isize = N-1 < 4 keeps nb low forever */
    const long isize = (long)n - 1;       /* deflate.py:605 */
    bitw w = {out, cap, 0, 0, 0, 0};
    size_t nt = 0;
    /* CSTATIC cur_cstatic 0..2 (deflate.py:746-762): 78 9C, then put(0x3, 3) */
    emit_byte(&w, 0, 0x78);
    emit_byte(&w, 1, 0x9c);
    w.dout = 2;
    put(&w, 0x3, 3);
    uint32_t adler1 = 1, adler2 = 0; /* deflate.py:749-751 */
    long di = 0;
    while (di <= isize) { /* deflate.py:771: the end sequence starts when di > isize */
        /* CSTATIC data step (deflate.py:823-834): Adler over x[di] */
        adler1 = (adler1 + x[di]) % 65521u;
        adler2 = (adler2 + adler1) % 65521u;
        /* SEARCH (deflate.py:975-994): cur_search = di-1 >= 0 and di < isize-3 */
        int found = 0;
        long dist = 0;
        if (di - 1 >= 0 && di < isize - 3) {
            for (long si = 0; si < cwindow; si++) { /* first set smatch bit = smallest distance */
                long q = di - si - 1;
                if (q < 0) break; /* deflate.py:989 */
                if (x[q] == x[di] && x[q + 1] == x[di + 1] && x[q + 2] == x[di + 2]) { /* matcher3 :407-413 */
                    found = 1;
                    dist = si + 1;
                    break;
                }
            }
        }
        if (!found) {
            /* literal (deflate.py:1005-1016) */
            put(&w, out_codes[x[di]], code_length[x[di]]);
            if (toks && nt < tok_cap) toks[nt] = (hdlz_token){(uint32_t)di, 0, x[di]};
            nt++;
            di += 1;
            continue;
        }
        /* SEARCHF (deflate.py:905-964): extend the nearest match */
        int match = 3;
        for (int k = 4; k <= maxmatch; k++) {
            if (di < isize - k && x[di - dist + k - 1] == x[di + k - 1]) match = k;
            else break;
        }
        /* DISTANCE (deflate.py:842-882) */
        int lencode = match + 254;
        put(&w, out_codes[lencode], code_length[lencode]);
        int ci = 0;
        while (copy_distance[ci + 1] <= dist) ci++; /* linear walk, deflate.py:858-882 */
        unsigned extra_dist = (unsigned)(dist - copy_distance[ci]);
        unsigned extra_bits = extra_distance_bits[ci / 2];
        unsigned outcode = rev_bits((unsigned)ci, 5) | (extra_dist << 5);
        if (extra_bits <= 4) {
            put(&w, outcode, 5 + extra_bits);
        } else { /* deflate.py:875-880 + :852-855: 8 bits, then the carry */
            put(&w, outcode & 0xFF, 8);
            put(&w, outcode >> 8, extra_bits - 3);
        }
        if (toks && nt < tok_cap) toks[nt] = (hdlz_token){(uint32_t)di, (uint16_t)match, (uint16_t)dist};
        nt++;
        /* CHECKSUM (deflate.py:888-897): Adler over the skipped bytes di+1 .. di+match-1 */
        for (long c = di + 1; c < di + match; c++) {
            adler1 = (adler1 + x[c]) % 65521u;
            adler2 = (adler2 + adler1) % 65521u;
        }
        di += match;
    }
    /* end sequence (deflate.py:771-819): EOB, pad, Adler-32 big-endian (s2 then s1) */
    put(&w, out_codes[256], code_length[256]);
    if (w.doo != 0) {
        emit_byte(&w, w.dout, w.ob1);
        w.dout++;
    }
    emit_byte(&w, w.dout++, adler2 >> 8);
    emit_byte(&w, w.dout++, adler2 & 0xFF);
    emit_byte(&w, w.dout++, adler1 >> 8);
    emit_byte(&w, w.dout++, adler1 & 0xFF);
    if (ntok) *ntok = nt;
    *out_len = w.dout; /* R9: final o_oprogress */
    return w.overflow ? HDLZ_E_OUT_CAPACITY : HDLZ_OK;
}

int hdlz_oracle_compress(const uint8_t* in, size_t n, int cwindow, int maxmatch, uint8_t* out, size_t out_cap,
                         size_t* out_len) {
    *out_len = 0;
    return compress_core(in, n, cwindow, maxmatch, out, out_cap, out_len, NULL, 0, NULL);
}

/* token list (pos,len,dist) for debugging device mismatches */
int hdlz_oracle_tokens(const uint8_t* in, size_t n, int cwindow, int maxmatch, uint32_t* pos, uint16_t* len,
                       uint16_t* dist, size_t cap, size_t* ntok) {
    size_t ol = 0;
    size_t bound = hdlz_oracle_out_bound(n);
    uint8_t* tmp = (uint8_t*)malloc(bound);
    hdlz_token* t = (hdlz_token*)malloc(sizeof(hdlz_token) * (n + 1));
    int rc = compress_core(in, n, cwindow, maxmatch, tmp, bound, &ol, t, n + 1, ntok);
    if (rc == HDLZ_OK)
        for (size_t i = 0; i < *ntok && i < cap; i++) {
            pos[i] = t[i].pos;
            len[i] = t[i].len;
            dist[i] = t[i].dist;
        }
    free(t);
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------ inflate (STARTD) */
typedef struct {
    const uint8_t* z;
    long zn;
    long di;      /* byte index   */
    unsigned dio; /* bit in byte  */
} bitr;

static uint32_t b41(const bitr* r) { /* deflate.py:348: bytes di..di+3, little endian; zeros past the end */
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) {
        long a = r->di + k;
        if (a >= 0 && a < r->zn) v |= (uint32_t)r->z[a] << (8 * k);
    }
    return v;
}
static unsigned get4(const bitr* r, unsigned boffset, unsigned width) { /* deflate.py:517-519 */
    if (width == 0) return 0;
    uint64_t v = b41(r);
    return (unsigned)((v >> (r->dio + boffset)) & ((1u << width) - 1));
}
static void adv(bitr* r, unsigned width) { /* deflate.py:521-533 */
    unsigned t = r->dio + width;
    r->di += t >> 3;
    r->dio = t & 7;
}

/* ---- dynamic trees (BTYPE=2): deflate.py:1084-1202 (BL/READBL/REPEAT/INIT3/DISTTREE),
 * :1204-1400 (HF1..HF4/SPREAD: canonical code construction), :1447-1517 (D_NEXT/D_NEXT_2).
 * The reference builds instant tables with an incremental-mask retry for long codes; a canonical
 * count/offset decoder yields the same symbols for every VALID code.  For code descriptions that are
 * not valid prefix codes the reference's behaviour is undefined table garbage; here they are errors
 * (HDLZ_E_BAD_TREE), using zlib's acceptance rules: over-subscribed sets are rejected, incomplete sets
 * only allowed when they hold a single code. */
static const uint8_t code_length_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15}; /* :97-98 */

typedef struct {
    uint16_t count[16];   /* number of codes of each length */
    uint16_t symbol[320]; /* symbols ordered by (length, value) */
} canon;

/* returns 0 for a complete code, >0 incomplete (unused code space), <0 over-subscribed */
static int canon_build(canon* h, const uint8_t* len, int n) {
    uint16_t offs[16];
    for (int l = 0; l < 16; l++) h->count[l] = 0;
    for (int s = 0; s < n; s++) h->count[len[s]]++;
    int left = 1;
    for (int l = 1; l < 16; l++) {
        left <<= 1;
        left -= h->count[l];
        if (left < 0) return left;
    }
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + h->count[l]);
    for (int s = 0; s < n; s++)
        if (len[s]) h->symbol[offs[len[s]]++] = (uint16_t)s;
    return left;
}

/* decode one symbol (codes are packed MSB first, deflate.py:1295-1316 reverses them into the tables);
 * returns the symbol or -1 when no code of length <= 15 matches */
static int canon_decode(bitr* r, const canon* h) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
        code |= (int)get4(r, 0, 1);
        adv(r, 1);
        int count = h->count[l];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

int hdlz_oracle_inflate(const uint8_t* z, size_t zn, unsigned flags, uint32_t obsize, uint8_t* out,
                        size_t out_cap, size_t* out_len) {
    build_tables();
    *out_len = 0;
    /* obsize != 0 = reference-exact OBSIZE build: the stored-block LEN register is LOBSIZE bits wide
     * (deflate.py:329 `length = modbv()[LOBSIZE:]`, assigned at :714), so LEN is taken mod 2^LOBSIZE.
     * obsize == 0 = RFC behaviour: full 16-bit LEN, 32 KiB history. */
    uint32_t len_mask = 0xFFFFu;
    if (obsize != 0) {
        unsigned lob = 0;
        while ((2u << lob) <= obsize) lob++;
        len_mask = (1u << lob) - 1u;
    } else {
        obsize = 32768;
    }
    const long isize = (long)zn - 1; /* deflate.py:605 */
    bitr r = {z, (long)zn, 2, 0};   /* D0: di = 2 skips the zlib header unvalidated (deflate.py:644) */
    size_t dout = 0;
    int final = 0;
    if (zn < 5) return HDLZ_E_SHORT_INPUT; /* nb never rises (deflate.py:429-431,662) */
    for (;;) {
        /* HEADER (deflate.py:677-732) */
        final = (int)get4(&r, 0, 1);
        if (flags & HDLZ_INFLATE_ONEBLOCK) final = 1; /* ONEBLOCK build: BFINAL is not read (deflate.py:678), the first
                                                        * EOB (:1542) / end of the first stored block (:1617) ends the stream */
        unsigned hm = (flags & HDLZ_INFLATE_ASSUME_FIXED) ? 1u : get4(&r, 1, 2);
        if (hm == 3) return HDLZ_E_BAD_BTYPE;
        if (hm == 0) {
            /* stored: deflate.py:709-717 then COPY :1603-1626 */
            unsigned skip = 8 - r.dio;
            if (skip <= 2) skip = 16 - r.dio;
            unsigned length = get4(&r, skip, 16) & len_mask;
            adv(&r, skip + 16); /* now at NLEN (unchecked, D2); data is at di+2 */
            for (unsigned i = 0; i < length; i++) {
                if (r.di >= isize - 2) return HDLZ_E_NO_EOF; /* the reference would hold forever (:1600) */
                if (dout >= out_cap) return HDLZ_E_OUT_CAPACITY;
                out[dout++] = z[r.di + 2]; /* obyte = b3 */
                r.di += 1;
            }
            if (r.di >= isize - 2) return HDLZ_E_NO_EOF;
            if (!final) {
                r.di += 2;
                continue;
            }
            break;
        }
        adv(&r, 3);
        canon lencode, distcode;
        if (hm == 2) {
            /* BL (deflate.py:1090-1114): HLIT, HDIST, HCLEN, then the code-length code lengths */
            int nlen = (int)get4(&r, 0, 5) + 257;
            int ndist = (int)get4(&r, 5, 5) + 1;
            int ncode = (int)get4(&r, 10, 4) + 4;
            adv(&r, 14);
            if (nlen > 286 || ndist > 30) return HDLZ_E_BAD_TREE;
            uint8_t lengths[320];
            memset(lengths, 0, sizeof(lengths));
            for (int i = 0; i < ncode; i++) {
                lengths[code_length_order[i]] = (uint8_t)get4(&r, 0, 3);
                adv(&r, 3);
            }
            if (canon_build(&lencode, lengths, 19) != 0) return HDLZ_E_BAD_TREE;
            /* READBL / REPEAT (deflate.py:1116-1164, :1190-1202) */
            int idx = 0;
            while (idx < nlen + ndist) {
                int sym = canon_decode(&r, &lencode);
                if (sym < 0) return HDLZ_E_BAD_TREE;
                if (sym < 16) {
                    lengths[idx++] = (uint8_t)sym;
                } else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (idx == 0) return HDLZ_E_BAD_TREE;
                        prev = lengths[idx - 1];
                        rep = 3 + (int)get4(&r, 0, 2);
                        adv(&r, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)get4(&r, 0, 3);
                        adv(&r, 3);
                    } else {
                        rep = 11 + (int)get4(&r, 0, 7);
                        adv(&r, 7);
                    }
                    if (idx + rep > nlen + ndist) return HDLZ_E_BAD_TREE;
                    while (rep--) lengths[idx++] = (uint8_t)prev;
                }
            }
            if (lengths[256] == 0) return HDLZ_E_BAD_TREE; /* no end-of-block code */
            int err = canon_build(&lencode, lengths, nlen);
            if (err < 0 || (err > 0 && nlen - lencode.count[0] != 1)) return HDLZ_E_BAD_TREE;
            err = canon_build(&distcode, lengths + nlen, ndist);
            /* an empty distance set (literals only) is legal (RFC1951 3.2.7; zlib inflate_table max == 0, puff) */
            if (err < 0 || (err > 0 && ndist - distcode.count[0] > 1)) return HDLZ_E_BAD_TREE;
            if (r.di > isize - 3) return HDLZ_E_NO_EOF; /* header ran into the trailer / past the end */
        }
        /* NEXT / INFLATE loop */
        int eob = 0;
        while (!eob) {
            unsigned code;
            if (hm == 2) {
                int sym = canon_decode(&r, &lencode); /* NEXT with the dynamic leaves (deflate.py:1409-1445) */
                if (sym < 0) return HDLZ_E_BAD_SYMBOL;
                code = (unsigned)sym;
            } else {
                unsigned cto = get4(&r, 0, 9);            /* deflate.py:1411 */
                unsigned leaf = stat_leaves[cto & 511];  /* :1417 */
                /* the zero leaf (index 483 = symbol 287's code followed by a 1) exists in the DYNAMIC=False build's stat_leaves only:
                 * a DYNAMIC=True build decodes a fixed block through leaves BUILT from the fixed lengths (deflate.py:1066-1073, then
                 * HF1..SPREAD), where 287 is an ordinary 8-bit leaf -- executed reference: tests/golden/inflate_r3_vectors.json */
                if ((cto & 511) == 483 && !(flags & HDLZ_INFLATE_ASSUME_FIXED)) leaf = (287u << 4) | 8u;
                unsigned nbits = leaf & 15;
                code = leaf >> 4;
                if (nbits < 1) return HDLZ_E_BAD_SYMBOL; /* :1437-1439 */
                adv(&r, nbits);
            }
            /* INFLATE (deflate.py:1519-1591) */
            if (r.di > isize - 3) return HDLZ_E_NO_EOF; /* :1535-1539 */
            if (code == 256) {
                eob = 1;
            } else if (code < 256) {
                if (dout >= out_cap) return HDLZ_E_OUT_CAPACITY;
                out[dout++] = (uint8_t)code;
            } else {
                unsigned token = code - 257;
                if (token >= 29) return HDLZ_E_BAD_SYMBOL; /* CopyLength has 29 entries */
                unsigned el = extra_length_bits[token];
                unsigned tlength = copy_length[token] + get4(&r, 0, el);
                unsigned dc, more, distance;
                if (hm == 2) {
                    /* D_NEXT / D_NEXT_2 (deflate.py:1447-1517): extra length bits, distance symbol, extra bits */
                    adv(&r, el);
                    int ds = canon_decode(&r, &distcode);
                    if (ds < 0) return HDLZ_E_BAD_SYMBOL;
                    dc = (unsigned)ds;
                    if (dc >= 30) return HDLZ_E_BAD_DISTANCE;
                    more = extra_distance_bits[dc >> 1];
                    distance = copy_distance[dc] + get4(&r, 0, more);
                    adv(&r, more);
                } else {
                    unsigned t = get4(&r, el, 5);
                    dc = rev_bits(t, 5);
                    if (dc >= 30) return HDLZ_E_BAD_DISTANCE; /* CopyDistance has 30 entries */
                    more = extra_distance_bits[dc >> 1];
                    distance = copy_distance[dc] + get4(&r, el + 5, more);
                    adv(&r, el + 5 + more);
                }
                if (distance > dout || distance > obsize) return HDLZ_E_BAD_DISTANCE; /* :1506-1508, D8 */
                if (r.di >= isize - 2) return HDLZ_E_NO_EOF; /* COPY would hold forever (:1600) */
                if (dout + tlength > out_cap) return HDLZ_E_OUT_CAPACITY;
                for (unsigned i = 0; i < tlength; i++, dout++) out[dout] = out[dout - distance]; /* COPY :1627-1659 */
            }
        }
        if (final) break; /* D6 */
    }
    *out_len = dout;
    return HDLZ_OK;
}

/* ------------------------------------------------------------------ threaded batch drivers
 * (cpu_baseline leg of bench.py: one block per task, contiguous ranges per thread) */
typedef struct {
    const uint8_t* in;
    const uint64_t* in_off;
    size_t b0, b1;
    int cwindow, maxmatch;
    unsigned flags;
    uint8_t* out;
    size_t out_pitch;
    uint32_t* out_len;
    uint32_t* status;
    int inflate;
} batch_job;

static void* batch_worker(void* p) {
    batch_job* j = (batch_job*)p;
    for (size_t b = j->b0; b < j->b1; b++) {
        size_t ol = 0;
        const uint8_t* src = j->in + j->in_off[b];
        size_t n = (size_t)(j->in_off[b + 1] - j->in_off[b]);
        int rc = j->inflate ? hdlz_oracle_inflate(src, n, j->flags, 0, j->out + b * j->out_pitch, j->out_pitch, &ol)
                            : hdlz_oracle_compress(src, n, j->cwindow, j->maxmatch, j->out + b * j->out_pitch,
                                                   j->out_pitch, &ol);
        j->out_len[b] = (uint32_t)ol;
        j->status[b] = (uint32_t)rc;
    }
    return NULL;
}

static int run_batch(batch_job proto, size_t nblocks, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > nblocks) nthreads = (int)(nblocks ? nblocks : 1);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    batch_job* jobs = (batch_job*)malloc(sizeof(batch_job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto;
        jobs[t].b0 = nblocks * t / nthreads;
        jobs[t].b1 = nblocks * (t + 1) / nthreads;
        if (nthreads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    if (nthreads > 1)
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
    return 0;
}

int hdlz_oracle_compress_batch(const uint8_t* in, const uint64_t* in_off, size_t nblocks, int cwindow, int maxmatch,
                               uint8_t* out, size_t out_pitch, uint32_t* out_len, uint32_t* status, int nthreads) {
    build_tables();
    batch_job p = {in, in_off, 0, 0, cwindow, maxmatch, 0, out, out_pitch, out_len, status, 0};
    return run_batch(p, nblocks, nthreads);
}

int hdlz_oracle_inflate_batch(const uint8_t* in, const uint64_t* in_off, size_t nblocks, unsigned flags, uint8_t* out,
                              size_t out_pitch, uint32_t* out_len, uint32_t* status, int nthreads) {
    build_tables();
    batch_job p = {in, in_off, 0, 0, 0, 0, flags, out, out_pitch, out_len, status, 1};
    return run_batch(p, nblocks, nthreads);
}

/* table access for tests (checks against RFC1951 and the reference's literal tables) */
const uint16_t* hdlz_oracle_out_codes(void) { build_tables(); return out_codes; }
const uint16_t* hdlz_oracle_stat_leaves(void) { build_tables(); return stat_leaves; }
