/*
 * hdlz.h -- C-ABI of the MI355X-native HDL-deflate engine (libhdlz.so).
 *
 * This is the drop-in boundary for the one hot path of tomtor/HDL-deflate: everything that
 * happens between STARTC/STARTD and o_done inside the reference's `deflate(...)` block
 * (/root/reference/deflate.py:219-221 port list, :607-1664 engine).  The reference moves one
 * byte per clock through i_data/o_byte (deflate.py:599-605); here whole batches of independent
 * blocks are handed over as device buffers.  The Python port-protocol adapter
 * (hdl_deflate_amd/port.py) re-creates the IDLE/WRITE/READ/STARTC/STARTD surface on top of
 * these entry points; INTEGRATION.md shows the binding a reference maintainer would add and
 * keeps the history of this ABI (what changed in which version, measured crossovers).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM); the library never frees or retains caller
 *     memory; all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream);
 *   - OWNERSHIP (the DUT owns only its own RAMs, deflate.py:229-230, :275-286): the caller
 *     allocates every device buffer, scratch included -- the entry points that need scratch take it as
 *     (d_work, work_bytes) next to a hdlz_*_work_bytes() query and allocate NOTHING, so they can be
 *     captured into a HIP graph.  hdlz_inflate_batch / hdlz_archive_batch without d_work are conveniences
 *     that draw the same scratch from a library-owned stream-ordered pool; NOT inside a graph capture
 *     (they return HDLZ_E_BAD_PARAM there: graph memory nodes are unreliable on ROCm 7.2,
 *     tools/repro/graph_scratch.hip);
 *   - block b of a batch is d_in[in_off[b] .. in_off[b+1]) when d_in_off != NULL, otherwise
 *     d_in[b*in_pitch .. b*in_pitch + in_len); with d_in_off, in_len is an optional UPPER BOUND on the
 *     block lengths (0 = not stated), see the two batch calls;
 *   - output of block b goes to d_out + b*out_pitch; out_pitch % 4 == 0 and d_out 4-byte
 *     aligned (16 recommended); d_out_len[b] receives the byte count (the reference's final
 *     o_oprogress, deflate.py:814 / :1554), d_status[b] one of HDLZ_OK / HDLZ_E_*;
 *   - the return value is HDLZ_OK or HDLZ_E_BAD_PARAM / HDLZ_E_HIP for host-side failures;
 *     per-block failures are only reported through d_status.
 * There is no CPU implementation behind this ABI: without a gfx950 device every compute entry
 * point returns HDLZ_E_HIP.
 */
#ifndef HDLZ_H
#define HDLZ_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDLZ_VERSION 0x000600   /* history: INTEGRATION.md 4 */

/* command codes of the reference port surface (deflate.py:18) -- used by the adapter */
enum { HDLZ_IDLE = 0, HDLZ_WRITE = 1, HDLZ_READ = 2, HDLZ_STARTC = 3, HDLZ_STARTD = 4 };

/* per-block status.  The reference has no error port: where it raises a Python `Error`
 * (deflate.py, 22 sites) or stalls forever, this engine reports a status instead. */
enum {
    HDLZ_OK = 0,
    HDLZ_E_SHORT_INPUT = 1,         /* N < 5: reference never starts (deflate.py:429-431, :740-741; README:194) */
    HDLZ_E_OUT_CAPACITY = 2,        /* compress: out_pitch < hdlz_out_bound(n); inflate: output exceeds out_pitch */
    HDLZ_E_BAD_BTYPE = 3,           /* "Bad method" (deflate.py:719-721) */
    HDLZ_E_BAD_DISTANCE = 4,        /* distance code 30/31, distance > bytes produced or > obsize (deflate.py:1506-1508, :1581) */
    HDLZ_E_NO_EOF = 5,              /* "NO EOF!" (deflate.py:1535-1539) or input exhausted (:1600-1602 would stall) */
    HDLZ_E_DYNAMIC_UNSUPPORTED = 6, /* internal hand-over mark between the inflate passes; never returned */
    HDLZ_E_BAD_SYMBOL = 7,          /* literal/length symbol 286/287 ("< 1 bits", deflate.py:1437-1439) */
    HDLZ_E_BAD_PARAM = 8,
    HDLZ_E_HIP = 9,                 /* HIP runtime error / no device; see hdlz_last_error() */
    HDLZ_E_BAD_TREE = 10            /* dynamic block header does not describe a valid prefix code (the reference
                                       builds garbage tables there, deflate.py:1204-1400; zlib's rules are used) */
};

/* ---- inflate flags (semantics) */
#define HDLZ_INFLATE_ASSUME_FIXED 1u     /* DYNAMIC=False build: every block is decoded as BTYPE=1 (deflate.py:724-732) */
#define HDLZ_INFLATE_ONEBLOCK 8u         /* ONEBLOCK=True build (deflate.py:40-49): BFINAL is not read, the stream ends with its FIRST block */
/* ---- inflate flags (mapping hints: results are identical; at most one; none = chosen from the batch's shape) */
#define HDLZ_INFLATE_LANE_PER_STREAM 2u  /* 64 streams per wave (k_inflate_tok) + a second pass for streams with dynamic-tree blocks */
#define HDLZ_INFLATE_WAVE_PER_STREAM 4u  /* one wave per stream (k_inflate_dyn), any block type */
#define HDLZ_INFLATE_GROUP_PER_STREAM 64u /* 16 lanes per stream, history in LDS (k_inflate_grp) + the second pass */
/* hint for the whole-GPU path of large streams (results are identical): the streams are single fixed-Huffman blocks -- what STARTC
 * writes --, so only that chain of kernels is launched; without it the chain for any block types is launched beside it (a stream takes
 * one of the two, decided on the device; the other one's launches return at once and cost a stream of STARTC ~8 %).  A stream of
 * other block types given with this hint is decoded by the serial pass. */
#define HDLZ_INFLATE_ONE_FIXED_BLOCK 128u
/* ---- the shapes the default mapping switches at (measured crossovers; provenance: INTEGRATION.md 4) */
#define HDLZ_INFLATE_WAVE_THRESHOLD 22528u   /* up to this many streams: a wave per stream */
#define HDLZ_INFLATE_DYN_LANE_MIN 28672u     /* second pass: a lane per stream from this many flagged streams on, else a wave each */
#define HDLZ_INFLATE_BIN_MIN 64u             /* lane mapping, ragged input of more streams than this: lanes take the streams ordered by length class */
#define HDLZ_INFLATE_GROUP_MIN 8192u         /* 16 lanes per stream for batches of GROUP_MIN .. GROUP_MAX streams */
#define HDLZ_INFLATE_GROUP_MAX 16384u
#ifndef HDLZ_INFLATE_PAR_MIN                 /* (A/B builds override it) */
#define HDLZ_INFLATE_PAR_MIN 2048u           /* whole-GPU path (k_par_*): streams of at least this many bytes (in_len; ragged: the stated bound) ... */
#endif
#define HDLZ_INFLATE_PAR_LONG 16384u         /* ... up to PAR_BATCH_MAX of them when in_len >= PAR_LONG, else up to PAR_BATCH_SHORT_MAX */
#define HDLZ_INFLATE_PAR_BATCH_MAX 4096u
#define HDLZ_INFLATE_PAR_BATCH_SHORT_MAX 1024u

int hdlz_version(void);
const char* hdlz_status_string(int status);
const char* hdlz_last_error(void);

/* number of visible HIP devices with a gfx950 agent; 0 if none (then nothing below can run) */
int hdlz_device_count(void);

/* Worst-case compressed size of an n-byte block: 2 header bytes + 3 block-header bits + 9 bits
 * per literal + 7 EOB bits, padded, + 4 Adler bytes = 6 + ceil((9n+10)/8)  (SURVEY 8(a)). */
size_t hdlz_out_bound(size_t n);

/* The library-owned scratch pool behind the entry points WITHOUT d_work keeps up to 256 MiB cached between calls and returns
 * anything above that to the device when the stream synchronises; this gives the cached rest of the CURRENT device back as well. */
int hdlz_release_scratch(void);

/*
 * STARTC for a batch: zlib stream 78 9C, ONE final fixed-Huffman block, LZ77 with a `cwindow`
 * byte look-back, nearest 3-byte match extended to at most `maxmatch` bytes, greedy parse,
 * Adler-32 trailer.  Replaces deflate.py:616-633 (IDLE/STARTC), :1064-1082 (STATIC),
 * :734-834 (CSTATIC), :966-1016 (SEARCH), :899-964 (SEARCHF), :1018-1062 (SEARCH10),
 * :836-882 (DISTANCE), :884-897 (CHECKSUM), :535-567 (put/do_flush), :407-421 (matcher3),
 * :423-515 (fill_buf).  Output is bit-identical to the reference for the same
 * (bytes, CWINDOW, MATCH10): cwindow in [1,256] (reference builds: 32 FAST/LOWLUT, 256
 * otherwise, deflate.py:56-59), maxmatch 10 (MATCH10=True) or 5 (deflate.py:34-35).
 * Ragged input: a stated bound in_len <= 1024 lets the call pack several small blocks per wave; a block longer
 * than a stated bound gets HDLZ_E_BAD_PARAM in its status word.  Needs no scratch.
 */
int hdlz_compress_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                        uint64_t nblocks, int cwindow, int maxmatch, uint8_t* d_out, uint64_t out_pitch,
                        uint32_t* d_out_len, uint32_t* d_status, void* stream);

/*
 * STARTD for a batch of independent zlib streams: 2 header bytes skipped unvalidated, blocks
 * until BFINAL -- stored (BTYPE 0), fixed-Huffman (BTYPE 1) and dynamic-tree (BTYPE 2, deflate.py:1084-1517) --,
 * 4 trailer bytes required but Adler-32 not verified: exactly the reference's acceptance (deflate.py:635-651
 * IDLE/STARTD, :656-732 HEADER, :1402-1445 NEXT, :1519-1591 INFLATE, :1593-1659 COPY, :517-533 get4/adv).
 * `obsize` != 0 selects the reference-exact behaviour of an OBSIZE build (deflate.py:61-62):
 * back-references may reach at most obsize bytes and a stored block's LEN is taken modulo
 * 2^floor(log2(obsize)) because the reference's `length` register is LOBSIZE bits wide
 * (deflate.py:329, :714).  obsize == 0 = RFC1951 behaviour (32 KiB history, 16-bit LEN).
 * Ragged input: in_len, if not 0, is the caller's UPPER BOUND on the stream lengths (the whole-GPU path sizes its pieces, its grids AND
 * its scratch from it -- per stream, whatever the streams' real lengths: state a TIGHT bound, or 0 for batches of many short streams;
 * a stream longer than the bound is still decoded, by the serial pass).
 *
 * hdlz_inflate_batch_ws: the call with CALLER-OWNED scratch.  d_work: device memory, 256-byte aligned, work_bytes long, used only
 * during the call's own launches (stream-ordered: it may be reused by the next call on the same stream).
 * hdlz_inflate_work_bytes(nstreams, in_len, out_pitch, flags, ragged) is what the fastest mapping of that shape uses; with LESS
 * (down to d_work = NULL) the call still succeeds with the same results through mappings that need less: the whole-GPU path runs the
 * streams in groups that fit or is skipped, the lane mapping takes the streams in index order, the second pass runs a wave per stream.
 * Nothing is allocated; every launch is capturable.  hdlz_inflate_batch is the same call with scratch from the library's pool.
 */
size_t hdlz_inflate_work_bytes(uint64_t nstreams, uint32_t in_len, uint64_t out_pitch, uint32_t flags, int ragged);
int hdlz_inflate_batch_ws(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                          uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                          uint32_t* d_out_len, uint32_t* d_status, void* d_work, size_t work_bytes, void* stream);
int hdlz_inflate_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                       uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                       uint32_t* d_out_len, uint32_t* d_status, void* stream);

/*
 * Archive compaction (no reference counterpart -- the reference drains its output
 * one byte per READ, deflate.py:601): copies d_len[b] bytes of row b (d_rows + b*row_pitch) to
 * d_archive + d_off[b].  d_off is the exclusive scan of the lengths, computed by the caller (across GPUs:
 * after the all-gather of the lengths).  Works for compress and inflate outputs alike; d_archive may be pinned host memory.
 */
int hdlz_compact_batch(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, const uint64_t* d_off,
                       uint64_t nblocks, uint8_t* d_archive, void* stream);

/*
 * The same gather with the scan inside: d_off[0 .. nblocks] is WRITTEN -- d_off[b] = sum of d_len[0 .. b), d_off[nblocks] =
 * the archive's length -- and row b is copied to d_archive + d_off[b], all in one launch (a ticketed decoupled look-back over tiles
 * of 256 rows).  d_off is at once the ragged-input index hdlz_inflate_batch / hdlz_compress_batch take (d_in_off).  archive_cap:
 * bytes writable at d_archive; rows that would end beyond it are not copied -- compare d_off[nblocks] with archive_cap after the
 * call (sum of row bounds = always enough).  d_archive must be device memory.  nblocks < 2^31.
 * _ws: d_work of at least hdlz_archive_work_bytes(nblocks) bytes (8 per tile of 256 rows), 8-byte aligned, REQUIRED (else
 * HDLZ_E_BAD_PARAM); nothing is allocated.  hdlz_archive_batch takes the same scratch from the library's pool.
 */
size_t hdlz_archive_work_bytes(uint64_t nblocks);
int hdlz_archive_batch_ws(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, uint64_t nblocks,
                          uint8_t* d_archive, uint64_t archive_cap, uint64_t* d_off, void* d_work, size_t work_bytes, void* stream);
int hdlz_archive_batch(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, uint64_t nblocks,
                       uint8_t* d_archive, uint64_t archive_cap, uint64_t* d_off, void* stream);

/* ---- one LARGE stream on the whole GPU ---------------------------------------------------------------------
 * Same STARTC semantics and bit-identical output as hdlz_compress_batch with nblocks = 1
 * (deflate.py:616-633 IDLE/STARTC ... :884-897 CHECKSUM: the reference handles one stream per START), but the
 * stream's 2 KiB tiles are spread over all compute units (parallel passes joined by scans).  Meant for
 * streams of >= 16 KiB (the measured crossover with one wave of the batch call is ~8 KiB) up to the reference's
 * LMAX range and beyond.
 *   d_in / in_len   the stream (in_len >= 5, else *d_status = HDLZ_E_SHORT_INPUT); readable up to in_len
 *                   rounded up to 16 bytes
 *   d_out / out_cap 4-byte aligned, out_cap >= hdlz_out_bound(in_len) rounded up to 4 (else HDLZ_E_OUT_CAPACITY)
 *   d_work          device scratch of hdlz_stream_work_bytes(in_len) bytes, 8-byte aligned
 * Returns HDLZ_OK when the launches were queued; *d_out_len, *d_status as in hdlz_compress_batch. */
size_t hdlz_stream_work_bytes(size_t in_len);

/* The same for nblocks blocks of in_len bytes each (block b at d_in + b*in_pitch -> d_out + b*out_pitch, one stream per
 * block as in hdlz_compress_batch): all tiles of all blocks share the passes.  For batches of a few to a few thousand
 * LARGE blocks (>= 256 KiB), where one wave per block (hdlz_compress_batch) leaves the GPU idle.  in_len >= 5 and
 * out_pitch >= hdlz_out_bound(in_len) rounded up to 4 are parameter errors here (not per-block statuses). */
size_t hdlz_streams_work_bytes(size_t in_len, uint64_t nblocks);
int hdlz_compress_streams(const uint8_t* d_in, uint64_t in_pitch, uint32_t in_len, uint64_t nblocks, int cwindow,
                          int maxmatch, uint8_t* d_out, uint64_t out_pitch, uint32_t* d_out_len, uint32_t* d_status,
                          void* d_work, size_t work_bytes, void* stream);
int hdlz_compress_stream(const uint8_t* d_in, uint32_t in_len, int cwindow, int maxmatch, uint8_t* d_out,
                         uint64_t out_cap, uint32_t* d_out_len, uint32_t* d_status, void* d_work,
                         size_t work_bytes, void* stream);

/* ---- STARTC for a stream that arrives in pieces (the streaming mode of the port) ------------------
 * The reference compresses WHILE the caller is still WRITE-ing (test_deflate.py:197-286): position di is encoded as soon as
 * ten more bytes are known (`di >= isize - 10 and i_mode != IDLE` stalls, deflate.py:768-770) and output becomes readable as
 * it is produced (put / do_flush, deflate.py:535-567).  hdlz_compress_chunk is that mode for a device-resident stream: any
 * number of calls produce ONE zlib stream / ONE deflate block, bit-identical to hdlz_compress_batch over the whole input.
 *   d_state   64-byte device-resident session (hdlz_cstate), ZEROED by the caller before the first call of a stream
 *   d_in      the stream from its first byte on (the pointer may change between calls, the bytes [0, in_len) may not),
 *             in_len = the bytes known so far
 *   q_end     this call encodes the positions [state.pos, q_end).  Not final: q_end <= in_len - 11 (the reference's stall
 *             margin: every encoded position then has its whole look-ahead, whatever the final length will be) and
 *             q_end - state.pos a positive multiple of 32.  Final: q_end == in_len = the stream length (>= 5, else
 *             HDLZ_E_SHORT_INPUT); writes EOB, padding, Adler-32 and sets done.
 *   d_out     the whole output stream, linear, 4-byte aligned; out_cap >= hdlz_out_bound(final length) + 2400
 * CONTRACT: the stream is the one the reference writes for an EAGER writer (test_deflate.py:250-258 -- its own harness -- keeps the
 * writer >= 21 bytes ahead), whatever the arrival pattern of the pieces.  The reference's bitstream depends on the writer's timing:
 * fill_buf latches iram[di+4 .. di+9] while the FSM stalls at di >= isize - 10 (deflate.py:466-500, :768-770), SEARCHF then compares
 * against those stale registers (deflate.py:913-952), and a writer that supplies a byte every 3rd .. 8th iteration gets one match cut
 * short (1262 instead of 1260 bytes on the recorded fixture, tests/golden/streaming_r3_vectors.json; both streams inflate to the
 * input).  That clock-by-clock dependence is NOT reproduced -- it would take a cycle-accurate model of the FSM on the host -- and is
 * recorded as fixtures instead.  Likewise the output memory: this engine never overwrites unread output.
 * After a call state.out_len complete output bytes are readable at d_out (the final call: the stream length, R9).
 * Violations are reported in state.status (HDLZ_E_BAD_PARAM / _OUT_CAPACITY / _SHORT_INPUT); a failed or finished session
 * ignores further calls.  One wave per call: the port adapter's path, not a throughput path. */
typedef struct hdlz_cstate {
    uint32_t pos;          /* positions [0, pos) are encoded */
    uint32_t skip;         /* positions from pos on that the last match already covers (greedy-parse state) */
    uint32_t out_words;    /* complete 32-bit words already in d_out */
    uint32_t base_bits;    /* valid bits of carry_word */
    uint32_t carry_word;   /* the partial output word (put's ob1/doo, deflate.py:535-560) */
    uint32_t adler_a;      /* sum x_p mod 65521 */
    uint32_t adler_c;      /* sum p * x_p mod 65521 */
    uint32_t started;
    uint32_t done;         /* the trailer is written; out_len is the length of the zlib stream */
    uint32_t out_len;      /* complete output bytes readable so far */
    uint32_t status;       /* HDLZ_OK or HDLZ_E_* */
    uint32_t reserved[5];
} hdlz_cstate;
int hdlz_compress_chunk(const uint8_t* d_in, uint32_t in_len, uint32_t q_end, int final, int cwindow, int maxmatch,
                        uint8_t* d_out, uint64_t out_cap, void* d_state, void* stream);

/* ---- STARTD for a stream that arrives in pieces -----------------------------------------------------------------------
 * The reference inflates while input is still being written and while the caller drains oram: it stalls on input
 * (`di >= isize - 4 and not i_mode == IDLE`, deflate.py:1529-1530; COPY: :1600-1602) and on output room
 * (`do >= i_raddr + OBSIZE`, deflate.py:1531-1534, :1597-1599).  hdlz_inflate_chunk is that mode: the decoder state lives in
 * a device-resident hdlz_istate (ZEROED by the caller before the first call) and every call decodes until
 *   - the stream ends                      -> state.done = 1, state.out_pos = the output length
 *   - the input known so far runs out      -> state.need = 1  (call again with more bytes / final = 1)
 *   - out_limit output bytes are reached   -> state.need = 2  (call again with a larger limit: the reader has advanced)
 *   - the stream is bad                    -> state.status = HDLZ_E_* (the codes of hdlz_inflate_batch; also
 *                                             HDLZ_E_OUT_CAPACITY when the stream needs more than out_cap bytes and
 *                                             out_limit >= out_cap: no larger limit could help)
 * always stopping BETWEEN two tokens, so the bytes [0, state.out_pos) of d_out are final after every call.
 *   d_in / in_len  the stream from its first byte on, in_len = bytes known so far (the pointer may change between calls,
 *                  the bytes may not); final != 0: in_len is the stream length and the reference's end-of-input checks apply
 *   flags          HDLZ_INFLATE_ASSUME_FIXED / HDLZ_INFLATE_ONEBLOCK;  obsize as in hdlz_inflate_batch
 *   d_out/out_cap  the whole output, linear (back-references read it);  out_limit: produce at most this many bytes in total
 * Results are identical to hdlz_inflate_batch on the complete stream.  One wave per call (the port adapter's path). */
typedef struct hdlz_istate {
    uint32_t bitpos;       /* next stream bit to decode */
    uint32_t out_pos;      /* output bytes produced (final, readable) */
    uint32_t phase;        /* 0 at a block header, 1 inside a Huffman block, 2 inside a stored block */
    uint32_t final_;       /* BFINAL of the current block */
    uint32_t hm;           /* BTYPE of the current block */
    uint32_t srem;         /* stored block: bytes still to copy */
    uint32_t nlen, ndist;  /* dynamic block: HLIT + 257, HDIST + 1 */
    uint32_t started, done, status;
    uint32_t need;         /* 0, 1 = more input, 2 = more output room */
    uint32_t reserved[4];
    uint8_t lengths[320];  /* dynamic block: the code lengths (the decode tables are rebuilt from them on resume) */
} hdlz_istate;
int hdlz_inflate_chunk(const uint8_t* d_in, uint32_t in_len, int final, uint32_t flags, uint32_t obsize, uint8_t* d_out,
                       uint64_t out_cap, uint32_t out_limit, void* d_state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HDLZ_H */
