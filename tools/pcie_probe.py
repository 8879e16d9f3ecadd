#!/usr/bin/env python3
"""Developer probe: host <-> device copy rates from page-locked memory (GB/s), one stream and several, both directions at once:
what bounds a path that moves compressed BAM bytes up and results down.   python tools/pcie_probe.py"""
import time

import torch

dev = torch.device("cuda:0")
n = 2 << 30
host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
devb = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]


def run(pairs, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for (dst, src), st in zip(pairs, streams):
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {3 * len(pairs) * n / dt / 1e9:.1f} GB/s")


run([(devb[0], host[0])], "host -> device, 1 stream")
run([(devb[i], host[i]) for i in range(4)], "host -> device, 4 streams")
run([(host[0], devb[0])], "device -> host, 1 stream")
run([(host[i], devb[i]) for i in range(4)], "device -> host, 4 streams")
run([(devb[0], host[0]), (devb[1], host[1]), (host[2], devb[2]), (host[3], devb[3])], "both directions, 2 + 2 streams (sum)")
