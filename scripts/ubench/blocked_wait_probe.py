"""Does a stream whose head is a not-yet-satisfied event wait slow the kernels of OTHER streams (command-processor artefact suspected
behind the fp32 leg's 87 -> 101 ms with one forward of run-ahead)?   python scripts/ubench/blocked_wait_probe.py
Stream C spins for ~60 ms and records E; stream B optionally waits for E (blocked all that time) and then runs one kernel; stream A runs
2000 small kernels meanwhile and is timed with HIP events."""
import sys, time
import torch
x = torch.zeros(1 << 14, device="cuda")
A, B, C = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
extra = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0)]
def run(blocked, n_wait=1):
    torch.cuda.synchronize()
    E = torch.cuda.Event()
    with torch.cuda.stream(C):
        torch.cuda._sleep(int(60e-3 * 2.0e9))
        E.record(C)
    waiters = [B] + extra[:n_wait - 1]
    if blocked:
        for w in waiters:
            with torch.cuda.stream(w):
                w.wait_event(E)
                x.add_(0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(A):
        s.record(A)
        for _ in range(2000): x.add_(1)
        e.record(A)
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 2000 * 1e3
for rep in range(3):
    print(f"us per small kernel on stream A:  nobody waiting {run(False):6.2f}   one stream blocked on an event {run(True):6.2f}"
          + (f"   {len(extra) + 1} streams blocked {run(True, len(extra) + 1):6.2f}" if extra else ""), flush=True)
