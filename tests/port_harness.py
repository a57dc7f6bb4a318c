"""The reference's streaming harness (test_deflate.py:115-195 STARTD leg / :197-286 STARTC leg) as ONE function over any
DUT with the ten ports and a cycle() -- the executed reference (oracle/gen_golden_r2.py wraps it in the stand-in kernel)
and hdl_deflate_amd.port.DeflatePort alike.  Two throttles extend it for the back-pressure fixtures:
  read_every  k: the reader takes a byte only every k-th loop iteration (a slow reader: the engine runs into the
              `do >= i_raddr + OBSIZE` hold, deflate.py:1531-1534)
  write_every k: the writer supplies a byte only every k-th iteration; in between the stale WRITE command stays on the
              ports, as in the harness's own "Wait for space" branch (test_deflate.py:167,258) -- driving IDLE there would
              tell the reference that the input has ended (deflate.py:768-770, :1529)
A trace of (cycle, bytes written, bytes read, o_iprogress, o_oprogress) is sampled every `trace_every` cycles."""

IDLE, WRITE, READ, STARTC, STARTD = range(5)


def stream_leg(dut, s, payload, start_cmd, maxw=32, short_input=False, read_every=1, write_every=1, trace_every=0,
               limit=None):
    """s: dict of the signals (i_mode, i_data, i_waddr, i_raddr, o_iprogress, o_oprogress, o_byte, o_done).
    Returns (bytes read, final o_oprogress, trace, stats)."""
    i_mode, i_waddr, i_raddr, i_data = s["i_mode"], s["i_waddr"], s["i_raddr"], s["i_data"]
    o_oprogress, o_iprogress, o_byte, o_done = s["o_oprogress"], s["o_iprogress"], s["o_byte"], s["o_done"]
    cycles = [0]

    def tick():
        dut.cycle()
        cycles[0] += 1

    i_mode.next = WRITE          # CLEAR OLD INPUT
    i_waddr.next = 0
    i_raddr.next = 0
    tick()
    i_mode.next = start_cmd
    tick()
    i = ri = it = 0
    res = bytearray()
    trace = []
    stats = {"max_ahead_of_reader": 0, "max_writer_lead": 0, "wait": 0}
    n = len(payload)
    if limit is None:
        limit = 2000 * (n + 100) * max(read_every, write_every)      # far beyond any legitimate run: a hang fails fast
    for _ in range(limit):
        it += 1
        if ri < int(o_oprogress) and it % read_every == 0:
            did_read = 1
            i_mode.next = READ
            i_raddr.next = ri
            tick()
            ri += 1
        else:
            did_read = 0
        if short_input and i == 0:
            i_mode.next = WRITE      # test_deflate.py:239-248 "SHORT INPUT": one WRITE of 0 at address 4
            i_waddr.next = 4
            i_data.next = 0
            i = 1
        elif not short_input and i < n:
            if int(o_iprogress) > i - maxw and it % write_every == 0:
                i_mode.next = WRITE
                i_waddr.next = i
                i_data.next = payload[i]
                i += 1
            else:
                stats["wait"] += 1   # the stale command stays on the ports
        else:
            i_mode.next = IDLE
        tick()
        if did_read:
            res.append(int(o_byte))
        stats["max_ahead_of_reader"] = max(stats["max_ahead_of_reader"], int(o_oprogress) - ri)
        stats["max_writer_lead"] = max(stats["max_writer_lead"], i - int(o_iprogress))
        if trace_every and cycles[0] % trace_every == 0:
            trace.append([cycles[0], i, ri, int(o_iprogress), int(o_oprogress)])
        if o_done and int(o_oprogress) == ri:
            break
    else:
        raise AssertionError("harness did not finish")
    i_mode.next = IDLE
    tick()
    stats["cycles"] = cycles[0]
    return bytes(res), int(o_oprogress), trace, stats
