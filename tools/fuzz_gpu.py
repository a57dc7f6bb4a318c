#!/usr/bin/env python3
"""GPU soak test (not part of pytest: runs for --seconds): random ragged batches, random CWINDOW /
MATCH10 / alphabets / alignments; every block of every batch is compared with the CPU oracle
(threaded), compress and inflate.  Exit code 1 on the first mismatch, with a reproducer line."""
import argparse
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, ".")
import torch                                   # noqa: E402
import hdl_deflate_amd                         # noqa: E402
from oracle import oracle as O                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    eng = hdl_deflate_amd.Engine()
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    it = blocks = nbytes = 0
    while time.time() < t_end:
        it += 1
        B = int(rng.integers(1, 3000))
        kind = int(rng.integers(0, 6))
        maxlen = int(rng.choice([16, 64, 300, 2048, 2100, 5000, 70000]))
        lens = rng.integers(0, maxlen + 1, size=B)
        if maxlen >= 5000:
            lens = lens[: max(1, B // 40)]
        B = len(lens)
        total = int(lens.sum())
        nsym = int(rng.choice([1, 2, 3, 4, 8, 26, 256]))
        if kind == 0:
            data = rng.integers(0, nsym, size=total, dtype=np.uint8) + (0 if nsym == 256 else int(rng.integers(0, 200)))
        elif kind == 1:   # periodic with noise
            per = int(rng.integers(1, 40))
            base = rng.integers(0, 256, size=per, dtype=np.uint8)
            data = np.tile(base, total // per + 1)[:total].copy()
            noise = rng.random(total) < 0.02
            data[noise] = rng.integers(0, 256, size=int(noise.sum()), dtype=np.uint8)
        elif kind == 2:   # runs
            data = np.repeat(rng.integers(0, 4, size=total // 3 + 1, dtype=np.uint8) + 65, 3)[:total].copy()
        elif kind == 5:   # a wide, skewed alphabet: compressible with 100..250 distinct symbols (dynamic blocks of many codes)
            data = np.minimum(255, np.abs(rng.normal(0, float(rng.choice([10, 40, 70, 120])), size=total))).astype(np.uint8)
        else:             # markov-ish text
            words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 9)), dtype=np.uint8)) for _ in range(50)]
            buf = b" ".join(words[int(i)] for i in rng.integers(0, 50, size=total // 3 + 2))
            data = np.frombuffer(buf[:total].ljust(total, b"x"), dtype=np.uint8).copy()
        mis = int(rng.integers(0, 16))
        flat = np.concatenate([np.zeros(mis, np.uint8), data.astype(np.uint8), np.zeros(64, np.uint8)])
        off = (np.concatenate([[0], np.cumsum(lens)]) + mis).astype(np.int64)
        cw = int(rng.choice([1, 2, 5, 16, 31, 32, 33, 48, 64, 65, 128, 255, 256]))
        mm = int(rng.choice([5, 10]))
        d_in = torch.from_numpy(flat).cuda()
        d_off = torch.from_numpy(off).cuda()
        out, ol, st = eng.compress_batch(d_in, in_off=d_off, cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        ho, hl, hs = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        ro, rl, rs = O.compress_batch(flat, off.astype(np.uint64), cw, mm, out_pitch=ho.shape[1], nthreads=16)
        mask = np.arange(ho.shape[1])[None, :] < rl[:, None]
        if not ((hs == rs).all() and (hl == rl).all() and ((ho == ro) | ~mask).all()):
            bad = int(np.nonzero((hs != rs) | (hl != rl) | (((ho != ro) & mask).any(1)))[0][0])
            print("COMPRESS MISMATCH it=%d seed=%d block=%d n=%d cw=%d mm=%d kind=%d mis=%d status gpu/ref %d/%d len %d/%d"
                  % (it, a.seed, bad, lens[bad], cw, mm, kind, mis, hs[bad], rs[bad], hl[bad], rl[bad]))
            return 1
        # the rows gathered into one archive (scan + gather in one launch): the streams back to back, offsets = the exclusive scan
        if it % 2 == 0:
            arc, aoff = eng.archive(out, ol)
            torch.cuda.synchronize()
            ha, hoff = arc.cpu().numpy(), aoff.cpu().numpy()
            want_off = np.concatenate([[0], np.cumsum(hl.astype(np.int64))])
            if not (hoff == want_off).all() or ha[: want_off[-1]].tobytes() != b"".join(ho[b, :hl[b]].tobytes() for b in range(B)):
                print("ARCHIVE MISMATCH it=%d seed=%d B=%d cw=%d mm=%d kind=%d" % (it, a.seed, B, cw, mm, kind))
                return 1
            if (hs == 0).all() and B >= 1:          # ... and inflated straight from it (ragged input = the archive's offsets)
                capa = (int(lens.max()) + 15) // 16 * 16 + 16
                ab, abl, abs_ = eng.inflate_batch(arc, in_off=aoff, out_pitch=capa, flags=int(rng.choice([0, 0, 2, 4, 64, 128])),
                                                  in_len=[None, int(hl.max()), max(1, int(hl.max()) // 2)][int(rng.integers(0, 3))])   # (the caller's bound on the lengths: none / right / too small)
                torch.cuda.synchronize()
                hb, hbl, hbs = ab.cpu().numpy(), abl.cpu().numpy(), abs_.cpu().numpy()
                for b in range(min(B, 2000)):
                    if hbs[b] != 0 or hbl[b] != lens[b] or hb[b, :lens[b]].tobytes() != flat[off[b]:off[b + 1]].tobytes():
                        print("ARCHIVE-INFLATE MISMATCH it=%d seed=%d block=%d n=%d cw=%d mm=%d status %d len %d" %
                              (it, a.seed, b, lens[b], cw, mm, hbs[b], hbl[b]))
                        return 1
        # inflate the compressed rows back on the GPU (padded rows, fixed pitch) for blocks that compressed
        okb = hs == 0
        if okb.any():
            cap = (int(lens.max()) + 15) // 16 * 16 + 16
            back, bl, bs = eng.inflate_batch(out, out_pitch=cap, flags=int(rng.choice([0, 2, 2, 4, 128, 64, 64])))   # default / lane / wave mapping
            torch.cuda.synchronize()
            hb, hbl, hbs = back.cpu().numpy(), bl.cpu().numpy(), bs.cpu().numpy()
            for b in np.nonzero(okb)[0][: 4000]:
                if hbs[b] != 0 or hbl[b] != lens[b] or hb[b, :lens[b]].tobytes() != flat[off[b]:off[b + 1]].tobytes():
                    print("INFLATE MISMATCH it=%d seed=%d block=%d n=%d cw=%d mm=%d status %d len %d" %
                          (it, a.seed, b, lens[b], cw, mm, hbs[b], hbl[b]))
                    return 1
            b0 = int(np.nonzero(okb)[0][0])
            assert zlib.decompress(ho[b0, :hl[b0]].tobytes()) == flat[off[b0]:off[b0 + 1]].tobytes()
        # our own streams damaged (one flipped bit, a cut tail, or untouched), under the reference's build variants -- DYNAMIC=False
        # (ASSUME_FIXED), ONEBLOCK, an OBSIZE build's history / LEN limits -- and every mapping: status and bytes against the oracle
        if it % 3 == 1 and okb.any():
            selb = np.nonzero(okb)[0][:300]
            zd = []
            for b in selb:
                zb_ = bytearray(ho[b, :hl[b]].tobytes())
                how_ = rng.random()
                if how_ < 0.6 and len(zb_) > 3:
                    zb_[int(rng.integers(2, len(zb_)))] ^= 1 << int(rng.integers(0, 8))
                elif how_ < 0.85:
                    zb_ = zb_[:max(1, len(zb_) - int(rng.integers(1, 7)))]
                zd.append(bytes(zb_))
            zoff = np.concatenate([[0], np.cumsum([len(z) for z in zd])]).astype(np.int64)
            zflat = np.frombuffer(b"".join(zd) + bytes(64), dtype=np.uint8).copy()
            fl_ = int(rng.choice([0, 2, 2, 4, 128, 64, 64])) | int(rng.choice([0, 1, 8, 9]))
            ob_ = int(rng.choice([0, 0, 512, 4096, 32768]))
            capd = (int(lens[selb].max()) + 300 + 15) // 16 * 16
            zb, zl, zst = eng.inflate_batch(torch.from_numpy(zflat).cuda(), in_off=torch.from_numpy(zoff).cuda(), out_pitch=capd, flags=fl_,
                                            obsize=ob_, in_len=[None, max(len(z) for z in zd), 64][int(rng.integers(0, 3))])
            torch.cuda.synchronize()
            hb, hbl, hbs = zb.cpu().numpy(), zl.cpu().numpy(), zst.cpu().numpy()
            for k, z in enumerate(zd):
                rc, ref = O.inflate(z, flags=fl_ & 9, obsize=ob_, out_cap=capd)
                if hbs[k] != rc or hb[k, :hbl[k]].tobytes() != ref:
                    print("DAMAGED-OWN-INFLATE MISMATCH it=%d seed=%d k=%d flags=%d obsize=%d status gpu/ref %d/%d len %d/%d z=%s" %
                          (it, a.seed, k, fl_, ob_, hbs[k], rc, hbl[k], len(ref), z.hex() if len(z) < 400 else "(long)"))
                    return 1
        # the whole flat buffer once more as ONE stream through the multi-wave path (needs >= 5 bytes)
        if total >= 5 and (cw <= 64 or total <= 300000):
            so, sl, ss = eng.compress_stream(d_in[mis:], total, cwindow=cw, maxmatch=mm)
            torch.cuda.synchronize()
            got = so[:int(sl.item())].cpu().numpy().tobytes()
            rc, ref = O.compress(flat[mis:mis + total].tobytes(), cw, mm)
            if int(ss.item()) != rc or got != ref:
                print("STREAM MISMATCH it=%d seed=%d n=%d cw=%d mm=%d kind=%d mis=%d status %d/%d len %d/%d"
                      % (it, a.seed, total, cw, mm, kind, mis, int(ss.item()), rc, len(got), len(ref)))
                return 1
        # stock zlib streams of the same blocks (dynamic / fixed / stored mix) through the two inflate passes
        if it % 4 == 0:
            sel = np.nonzero(lens >= 1)[0][:600]
            zs = [zlib.compress(flat[off[b]:off[b + 1]].tobytes(), int(rng.integers(0, 10))) for b in sel]
            if zs:
                zoff = np.concatenate([[0], np.cumsum([len(z) for z in zs])]).astype(np.int64)
                zflat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
                cap = (int(lens[sel].max()) + 15) // 16 * 16 + 16
                zb, zl, zst = eng.inflate_batch(torch.from_numpy(zflat).cuda(), in_off=torch.from_numpy(zoff).cuda(), out_pitch=cap,
                                                   flags=int(rng.choice([0, 2, 2, 4, 128, 64, 64])))
                torch.cuda.synchronize()
                hb, hbl, hbs = zb.cpu().numpy(), zl.cpu().numpy(), zst.cpu().numpy()
                for k, b in enumerate(sel):
                    if hbs[k] != 0 or hbl[k] != lens[b] or hb[k, :lens[b]].tobytes() != flat[off[b]:off[b + 1]].tobytes():
                        print("ZLIB-INFLATE MISMATCH it=%d seed=%d k=%d n=%d status %d len %d" % (it, a.seed, k, lens[b], hbs[k], hbl[k]))
                        return 1
                # the same streams damaged (one flipped bit, or cut short): status and bytes against the oracle, every mapping in turn
                zd = []
                for z in zs[:200]:
                    zb_ = bytearray(z)
                    if rng.random() < 0.7 and len(zb_) > 3:
                        zb_[int(rng.integers(2, len(zb_)))] ^= 1 << int(rng.integers(0, 8))
                    else:
                        zb_ = zb_[:max(1, len(zb_) - int(rng.integers(1, 7)))]
                    zd.append(bytes(zb_))
                zoff = np.concatenate([[0], np.cumsum([len(z) for z in zd])]).astype(np.int64)
                zflat = np.frombuffer(b"".join(zd) + bytes(64), dtype=np.uint8).copy()
                fl_ = int(rng.choice([0, 2, 2, 4, 128, 64, 64])) | (1 if rng.random() < 0.25 else 0)      # (now and then as the DYNAMIC=False build: every block decoded as fixed)
                zb, zl, zst = eng.inflate_batch(torch.from_numpy(zflat).cuda(), in_off=torch.from_numpy(zoff).cuda(), out_pitch=cap, flags=fl_)
                torch.cuda.synchronize()
                hb, hbl, hbs = zb.cpu().numpy(), zl.cpu().numpy(), zst.cpu().numpy()
                for k, z in enumerate(zd):
                    rc, ref = O.inflate(z, flags=fl_ & 1, out_cap=cap)
                    if hbs[k] != rc or hb[k, :hbl[k]].tobytes() != ref:
                        print("DAMAGED-ZLIB-INFLATE MISMATCH it=%d seed=%d k=%d flags=%d status gpu/ref %d/%d len %d/%d z=%s" %
                              (it, a.seed, k, fl_, hbs[k], rc, hbl[k], len(ref), z.hex() if len(z) < 400 else "(long)"))
                        return 1
        # now and then: a fixed-pitch batch of a few LARGE blocks cut from the same data (hdlz_compress_streams: all tiles of
        # all blocks share the stream passes), every block against the oracle
        if it % 8 == 5 and total >= 200000 and (cw <= 64 or total <= 400000):
            nbk = int(rng.integers(2, 12))
            nbytes_ = int(rng.integers(65536, max(65537, min(total // nbk, 700000))))
            if nbytes_ * nbk <= total and nbytes_ >= 65536:
                pitch_ = (nbytes_ + 15) // 16 * 16
                big = torch.zeros((nbk, pitch_), dtype=torch.uint8, device="cuda")
                big[:, :nbytes_] = d_in[mis:mis + nbk * nbytes_].view(nbk, nbytes_)
                bo, bl2, bs2 = eng.compress_batch(big, in_len=nbytes_, cwindow=cw, maxmatch=mm)
                torch.cuda.synchronize()
                hb2, hl2, hs2 = bo.cpu().numpy(), bl2.cpu().numpy(), bs2.cpu().numpy()
                foff = (np.arange(nbk + 1, dtype=np.uint64) * nbytes_)
                ro2, rl2, rs2 = O.compress_batch(flat[mis:mis + nbk * nbytes_], foff, cw, mm, out_pitch=hb2.shape[1], nthreads=16)
                mask2 = np.arange(hb2.shape[1])[None, :] < rl2[:, None]
                if not ((hs2 == rs2).all() and (hl2 == rl2).all() and ((hb2 == ro2) | ~mask2).all()):
                    print("STREAMS MISMATCH it=%d seed=%d nbk=%d n=%d cw=%d mm=%d kind=%d" % (it, a.seed, nbk, nbytes_, cw, mm, kind))
                    return 1
        # now and then: ONE large stream through STARTD (hdlz_inflate_par.hip: pieces, markers, pointer jumping -- or, for anything
        # that is not one valid fixed block, the device-side fallback to the serial decoder): our own stream of a slice of the data,
        # as it is, with one flipped bit, cut short, or with a tight output capacity -- status and bytes against the oracle
        if it % 6 == 2 and total >= 60000:
            sl_n = int(min(total, rng.integers(60000, 900000)))
            pad_ = torch.zeros((sl_n + 15) // 16 * 16 + 16, dtype=torch.uint8, device="cuda")
            pad_[:sl_n] = d_in[mis:mis + sl_n]
            so, sl_, ss_ = eng.compress_stream(pad_, sl_n, cwindow=cw, maxmatch=mm)
            zs_ = so[:int(sl_.item())].cpu().numpy().tobytes()
            how = int(rng.integers(0, 4))
            cap_ = sl_n + 64
            if how == 1 and len(zs_) > 8:
                zb_ = bytearray(zs_)
                zb_[int(rng.integers(2, len(zb_)))] ^= 1 << int(rng.integers(0, 8))
                zs_ = bytes(zb_)
            elif how == 2:
                zs_ = zs_[:int(rng.integers(len(zs_) // 2, len(zs_)))]
            elif how == 3:
                cap_ = max(16, (sl_n - int(rng.integers(0, 4096))) // 16 * 16)
            st_g, out_g = eng.inflate_bytes(zs_, out_cap=cap_)
            st_o, out_o = O.inflate(zs_, out_cap=(cap_ + 15) // 16 * 16)
            if st_g != st_o or out_g != out_o:
                print("SINGLE-STREAM INFLATE MISMATCH it=%d seed=%d n=%d how=%d cw=%d status %d / %d" % (it, a.seed, sl_n, how, cw, st_g, st_o))
                return 1
        # now and then: a slice of the data through the RESUMABLE kernels (hdlz_compress_chunk / hdlz_inflate_chunk), fed in
        # random pieces with random caps on the work per call -- must equal the one-shot oracle
        if it % 3 == 1 and total >= 5:
            sl_n = int(min(total, rng.integers(5, 60000)))
            piece = flat[mis:mis + sl_n].tobytes()
            rc, ref = O.compress(piece, cw, mm)
            cs = eng.compress_session(cwindow=cw, maxmatch=mm)
            i = 0
            while i < sl_n:
                k = int(rng.integers(1, 4000))
                cs.write(piece[i:i + k])
                i += k
                cs.step(max_positions=int(rng.choice([32, 96, 1024, 1 << 20])))
            guard = 0
            while not cs.done and guard < 100000:
                guard += 1
                if cs.step(final=True, max_positions=int(rng.choice([64, 2048, 1 << 20]))) != 0:
                    break
            if not cs.done or cs.output(0, cs.out_len) != ref:
                print("CHUNK-COMPRESS MISMATCH it=%d seed=%d n=%d cw=%d mm=%d" % (it, a.seed, sl_n, cw, mm))
                return 1
            z = zlib.compress(piece, int(rng.integers(0, 10))) if rng.random() < 0.7 else ref
            isn = eng.inflate_session()
            i, limit, guard = 0, 600, 0
            while not isn.done and guard < 200000:
                guard += 1
                if i < len(z):
                    k = int(rng.integers(1, 3000))
                    isn.write(z[i:i + k])
                    i += k
                if isn.step(final=(i >= len(z)), out_limit=limit) != 0:
                    break
                if isn.need == 2 or rng.random() < 0.3:
                    limit += int(rng.integers(1, 5000))
            if not isn.done or isn.output(0, isn.out_pos) != piece:
                print("CHUNK-INFLATE MISMATCH it=%d seed=%d n=%d zlen=%d" % (it, a.seed, sl_n, len(z)))
                return 1
        blocks += B
        nbytes += total
    print("fuzz OK: %d batches, %d blocks, %.1f MiB, seed %d" % (it, blocks, nbytes / 2 ** 20, a.seed))
    return 0


if __name__ == "__main__":
    sys.exit(main())
