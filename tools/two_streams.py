#!/usr/bin/env python3
"""Do two independent conv-stack forwards (the actor's and the target critic's view of next_obs in the critic phase)
finish sooner on two HIP streams than back to back on one?  Kernel boundaries of one chain could hide under the
other's work.  Prints the time of both orders."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curla_amd import ops  # noqa: E402

dev = "cuda"
B = 512
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
store = torch.randint(0, 256, (2048 * 84 * 84 * 9 + 32,), dtype=torch.uint8, device=dev)
ring = store[:2048 * 84 * 84 * 9].view(2048, 84, 84, 9)


def chain():
    idx = torch.randint(0, 2048, (B,), device=dev)
    h1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
    w1 = torch.randint(0, 8, (B,), device=dev, dtype=torch.int32)
    obs = ops.ObsRef.from_ring(ring, idx, h1, w1, B, (76, 76))
    w0, b = r(32, 9, 3, 3) * 0.1, r(32) * 0.1
    ws = [r(32, 32, 3, 3) * 0.1 for _ in range(3)]
    acts = [torch.empty(B, s, s, 32, device=dev) for s in (37, 35, 33, 31)]

    def run():
        ops.conv1_fwd(obs, w0, b, acts[0])
        for i in range(3):
            ops.conv_s1_fwd(acts[i], ws[i], b, acts[i + 1])
    return run


a, c = chain(), chain()
for _ in range(300):
    a()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n=30):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def serial():
    a()
    c()


def parallel():
    cur = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(cur)
    s1.wait_event(ev)
    s2.wait_event(ev)
    with torch.cuda.stream(s1):
        a()
    with torch.cuda.stream(s2):
        c()
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1)
    e2.record(s2)
    cur.wait_event(e1)
    cur.wait_event(e2)


print("two conv-stack forwards back to back on one stream: %.1f us" % timed(serial))
print("the same on two streams:                            %.1f us" % timed(parallel))
