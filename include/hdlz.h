/*
 * hdlz.h -- C-ABI of the MI355X-native HDL-deflate engine (libhdlz.so).
 *
 * This is the drop-in boundary for the one hot path of tomtor/HDL-deflate: everything that
 * happens between STARTC/STARTD and o_done inside the reference's `deflate(...)` block
 * (/root/reference/deflate.py:219-221 port list, :607-1664 engine).  The reference moves one
 * byte per clock through i_data/o_byte (deflate.py:599-605); here whole batches of independent
 * blocks are handed over as device buffers.  The Python port-protocol adapter
 * (hdl_deflate_amd/port.py) re-creates the IDLE/WRITE/READ/STARTC/STARTD surface on top of
 * these entry points; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM); the library never allocates, frees or
 *     retains caller memory; all work is enqueued asynchronously on `stream` (a hipStream_t
 *     passed as void*; NULL = the default stream);
 *   - block b of a batch is d_in[in_off[b] .. in_off[b+1]) when d_in_off != NULL, otherwise
 *     d_in[b*in_pitch .. b*in_pitch + in_len); with d_in_off, hdlz_compress_batch takes in_len as an optional
 *     upper bound on the block lengths (0 = unknown): a bound <= 1024 lets it pack several small blocks per wave
 *     (a block longer than a stated bound gets HDLZ_E_BAD_PARAM in its status word);
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

#define HDLZ_VERSION 0x000500   /* 0x000500, round 5: + a third inflate mapping, 16 lanes per stream with the stream's history in LDS (HDLZ_INFLATE_GROUP_PER_STREAM = 64,
                                   the default for batches of HDLZ_INFLATE_GROUP_MIN .. _MAX streams); new compress kernels (same bytes).  0x000401: same ABI, new inflate lane kernels (register-queue refill, second token group per round): the PMC records
                                   of profiles/traffic.json are tied to the version.  0x000400, round 4: + hdlz_release_scratch (bounded scratch pool); hdlz_compact_batch accepts a pinned-host destination; the opt-in
                                   two-phase inflate of 0x000301 (HDLZ_INFLATE_TWO_PHASE = 64) was measured slower than the one-pass kernel and is gone */

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
    HDLZ_E_DYNAMIC_UNSUPPORTED = 6, /* internal hand-over mark between the two inflate passes; never returned */
    HDLZ_E_BAD_SYMBOL = 7,          /* literal/length symbol 286/287 ("< 1 bits", deflate.py:1437-1439) */
    HDLZ_E_BAD_PARAM = 8,
    HDLZ_E_HIP = 9,                 /* HIP runtime error / no device; see hdlz_last_error() */
    HDLZ_E_BAD_TREE = 10            /* dynamic block header does not describe a valid prefix code (the reference
                                       builds garbage tables there, deflate.py:1204-1400; zlib's rules are used) */
};

/* inflate flags */
#define HDLZ_INFLATE_ASSUME_FIXED 1u /* DYNAMIC=False build: every block is decoded as BTYPE=1 (deflate.py:724-732) */
/* ONEBLOCK=True build (deflate.py:40-41; forced by LOWLUT, :43-49): BFINAL is never read (deflate.py:678) and the stream
 * ends at the end of its FIRST block -- EOB (deflate.py:1542) or the last stored byte (:1617) -- whatever follows */
#define HDLZ_INFLATE_ONEBLOCK 8u
/* mapping hints (results are identical): by default batches of at most HDLZ_INFLATE_WAVE_THRESHOLD streams are decoded
 * one wave per stream, larger ones one lane per stream, with a second pass for the streams that hold dynamic-tree blocks:
 * again one lane per stream when there are at least HDLZ_INFLATE_DYN_LANE_MIN of them (counted on the device), else one
 * wave per stream (measured crossovers on 2 KiB and 16 KiB streams, tools/bench_inflate_mapping.py) */
#define HDLZ_INFLATE_LANE_PER_STREAM 2u
#define HDLZ_INFLATE_WAVE_PER_STREAM 4u
#define HDLZ_INFLATE_WAVE_THRESHOLD 22528u
#define HDLZ_INFLATE_DYN_LANE_MIN 28672u
/* lane mapping, ragged input (d_in_off given) of more than this many streams: the lanes take the streams in the order of their
 * compressed-length class (a counting sort on the device, 4 bytes per stream of stream-ordered scratch; stream order if that
 * allocation fails) -- a wave runs as long as its longest stream, so streams of similar length share a wave.  Results are
 * identical; the output rows stay where their stream index puts them. */
#define HDLZ_INFLATE_BIN_MIN 64u
/* a batch of ONE stream of at least this many bytes (fixed-pitch form, no mapping hint) is cut into 1 KiB pieces and decoded by
 * the whole GPU when it is a single fixed-Huffman block -- the streams STARTC writes --, else by one wave as before (decided on
 * the device, same results); scratch: stream-ordered, 4 bytes per possible output byte (min(out_pitch, 172 * in_len); 8 up to round 4).
 * A batch of up to HDLZ_INFLATE_PAR_BATCH_MAX such streams (fixed pitch, no mapping hint) goes through the same path, every kernel
 * launched ONCE for all of them (round 5; one chain of launches per stream before): the batch kernels decode a stream as one serial
 * chain -- 5.9 ms for a 64 KiB stream however few there are -- so 256 streams of 64 KiB take 0.48 ms instead of 5.9, 256 of 1 MiB 4.4 ms
 * instead of 42; from ~8192 streams on the batch kernels win (profiles/r05_inflate_mapping.txt).  Scratch as above, per stream (more than 4 GiB: the batch goes through in groups of streams).
 * Streams shorter than HDLZ_INFLATE_PAR_LONG take the path in batches of up to HDLZ_INFLATE_PAR_BATCH_SHORT_MAX (the chain of launches costs
 * ~0.12 ms: one 8 KiB stream 0.12 instead of 0.75 ms, 1024 streams of 2 KiB 0.21 instead of 0.33 ms; 4096 of them: the wave mapping wins).
 * (The threshold was 16384 up to the first builds of 0x000500: the path cost 0.35 ms then.)
 * Ragged input (d_in_off given): in_len, if not 0, is the caller's UPPER BOUND on the stream lengths and the thresholds above apply to it (0 =
 * not stated: the batch kernels, as before); a stream longer than the bound is decoded by the serial pass -- right, only slower.
 * While `stream` is being captured into a HIP graph only ONE fixed-pitch stream of >= HDLZ_INFLATE_PAR_LONG bytes takes the path (what it
 * took before round 5; a batch goes to the batch kernels): same results, see hdlz_api.hip. */
#ifndef HDLZ_INFLATE_PAR_MIN          /* (A/B builds override it) */
#define HDLZ_INFLATE_PAR_MIN 2048u
#endif
#define HDLZ_INFLATE_PAR_LONG 16384u
#define HDLZ_INFLATE_PAR_BATCH_MAX 4096u
#define HDLZ_INFLATE_PAR_BATCH_SHORT_MAX 1024u
/* 16 lanes per stream (hdlz_inflate_grp.hip; round 5): the stream's history in a 2 KiB LDS ring, input and output in full lines, four
 * streams per wave -- the mapping for batches too small to fill the GPU one lane per stream and too large to give every stream a
 * wave: the default for HDLZ_INFLATE_GROUP_MIN <= nstreams <= HDLZ_INFLATE_GROUP_MAX (measured crossovers, tools/bench_inflate_mapping.py),
 * and for any batch when the flag is given.  Streams with dynamic-tree blocks take the usual second pass.  (The value 64 was the
 * two-phase inflate of 0x000301, rejected as unknown by 0x000400 / 0x000401.) */
#define HDLZ_INFLATE_GROUP_PER_STREAM 64u
#define HDLZ_INFLATE_GROUP_MIN 8192u
#define HDLZ_INFLATE_GROUP_MAX 16384u
/* lane-per-stream kernel variant (results are identical): the default and 16 = one token per round (k_inflate_tok),
 * 32 = one output byte per lockstep iteration (k_inflate, the round-1 kernel) */
#define HDLZ_INFLATE_TOKEN_ROUNDS 16u
#define HDLZ_INFLATE_BYTE_LOCKSTEP 32u

int hdlz_version(void);
const char* hdlz_status_string(int status);
const char* hdlz_last_error(void);

/* number of visible HIP devices with a gfx950 agent; 0 if none (then nothing below can run) */
int hdlz_device_count(void);

/* Worst-case compressed size of an n-byte block: 2 header bytes + 3 block-header bits + 9 bits
 * per literal + 7 EOB bits, padded, + 4 Adler bytes = 6 + ceil((9n+10)/8)  (SURVEY 8(a)). */
size_t hdlz_out_bound(size_t n);

/* The calls that need stream-ordered scratch (the dynamic-tree pass of the lane mapping, the parallel single-stream inflate) draw it
 * from the library's own per-device memory pool, which keeps up to 256 MiB cached between calls (a fresh device allocation per call
 * costs 10-40 ms) and returns anything above that to the device when the stream synchronises.  hdlz_release_scratch() gives the
 * cached rest of the CURRENT device back as well (e.g. before the caller's own large allocations).  HDLZ_OK / HDLZ_E_HIP. */
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
 */
int hdlz_compress_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                        uint64_t nblocks, int cwindow, int maxmatch, uint8_t* d_out, uint64_t out_pitch,
                        uint32_t* d_out_len, uint32_t* d_status, void* stream);

/*
 * STARTD for a batch of independent zlib streams: 2 header bytes skipped unvalidated, blocks
 * until BFINAL, stored (BTYPE 0), fixed-Huffman (BTYPE 1) and dynamic-tree (BTYPE 2, deflate.py:1084-1517;
 * handled by a second pass over the streams that hold such blocks -- see the mapping hints above; in the lane
 * mapping that pass runs in two stages, the second one for the few streams whose block codes more than 144
 * literal/length symbols, and keeps the list of its streams in stream-ordered scratch memory, hipMallocAsync /
 * hipFreeAsync on `stream`, 4 bytes per stream (if that allocation fails the wave mapping finishes the job); the
 * parallel path for ONE large stream or up to HDLZ_INFLATE_PAR_BATCH_MAX of them -- HDLZ_INFLATE_PAR_MIN below -- allocates 4 bytes per possible
 * output byte the same way, whatever the flags; no other case allocates, and every case stays capturable into a HIP
 * graph) blocks, 4 trailer bytes required
 * but Adler-32 not verified -- exactly the reference's acceptance (deflate.py:635-651 IDLE/STARTD,
 * :656-732 HEADER, :1402-1445 NEXT, :1519-1591 INFLATE, :1593-1659 COPY, :517-533 get4/adv).
 * `obsize` != 0 selects the reference-exact behaviour of an OBSIZE build (deflate.py:61-62):
 * back-references may reach at most obsize bytes and a stored block's LEN is taken modulo
 * 2^floor(log2(obsize)) because the reference's `length` register is LOBSIZE bits wide
 * (deflate.py:329, :714).  obsize == 0 = RFC1951 behaviour (32 KiB history, 16-bit LEN).
 */
int hdlz_inflate_batch(const uint8_t* d_in, const uint64_t* d_in_off, uint64_t in_pitch, uint32_t in_len,
                       uint64_t nstreams, uint32_t flags, uint32_t obsize, uint8_t* d_out, uint64_t out_pitch,
                       uint32_t* d_out_len, uint32_t* d_status, void* stream);

/*
 * Archive compaction (no reference counterpart -- the reference drains its output
 * one byte per READ, deflate.py:601): copies d_len[b] bytes of row b (d_rows + b*row_pitch) to
 * d_archive + d_off[b].  d_off is the exclusive scan of the lengths, computed by the caller (across GPUs:
 * after the all-gather of the lengths).  Works for compress and inflate outputs alike.
 */
int hdlz_compact_batch(const uint8_t* d_rows, uint64_t row_pitch, const uint32_t* d_len, const uint64_t* d_off,
                       uint64_t nblocks, uint8_t* d_archive, void* stream);

/*
 * The same gather with the scan inside (round 5): d_off[0 .. nblocks] is WRITTEN -- d_off[b] = sum of d_len[0 .. b), d_off[nblocks] =
 * the archive's length -- and row b is copied to d_archive + d_off[b], all in one launch (a ticketed decoupled look-back over tiles
 * of 256 rows; stream-ordered scratch: 8 bytes per tile).  d_off is at once the ragged-input index hdlz_inflate_batch / hdlz_compress_batch
 * take (d_in_off).  archive_cap: bytes writable at d_archive; rows that would end beyond it are not copied -- compare d_off[nblocks]
 * with archive_cap after the call (sum of row bounds = always enough).  d_archive must be device memory.  nblocks < 2^31.
 */
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
