#!/usr/bin/env python3
"""Does a HIP graph shorten the GPU time of a fixed kernel sequence?  Encoder forward at configs[1] shapes
(conv1 from the ring, 3 stride-1 convs, split-K fc GEMM, LayerNorm): eager launches vs graph replays."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import curla_amd
from curla_amd import ops

dev = torch.device("cuda")
aug = curla_amd.RandomCrop((84, 84), (76, 76))
agent = curla_amd.CurlSacAgent((9, 76, 76), (2,), dev, aug, hidden_dim=1024)
enc = agent.critic.encoder
B = 512
g = torch.Generator(device="cuda").manual_seed(0)
store = torch.randint(0, 256, (2048 * 84 * 84 * 9 + 32,), dtype=torch.uint8, device="cuda", generator=g)
ring = store[:2048 * 84 * 84 * 9].view(2048, 84, 84, 9)
idx = torch.randint(0, 2048, (B,), device="cuda", generator=g)
h1 = torch.randint(0, 9, (B,), device="cuda", generator=g).int()
w1 = torch.randint(0, 9, (B,), device="cuda", generator=g).int()
ref = ops.ObsRef.from_ring(ring, idx, h1, w1, B, (76, 76))
acts = enc.workspace(B, tag="graph").acts
z = torch.empty(B, enc.feature_dim, device="cuda")


def forward():
    enc.conv_forward(ref, acts)
    enc.head_forward(acts[-1].view(B, -1), z)


def timeit(fn, iters=300):
    for _ in range(600):  # clock warm-up
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


forward()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        for _ in range(4):  # four forwards per graph: 24 kernels
            forward()
t_eager = timeit(forward)
t_graph = timeit(graph.replay, iters=100) / 4
print(f"encoder forward (6 kernels): eager {t_eager:.1f} us, inside a graph {t_graph:.1f} us per forward")
