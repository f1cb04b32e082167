#!/usr/bin/env python3
"""Device-to-host copy bandwidth of this box by chunk size and number of concurrent streams (pinned destination buffers), and the
host's CBC-MAC rate per thread and for all threads: the two limits of the PCIe-inclusive garbling rate (commitment stage)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

dev = torch.device("cuda", 0)
src = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
src.random_(0, 255)
print("D2H, pinned destination; GB/s by chunk size x streams")
for chunk_mb in (4, 16, 64, 256, 1024):
    row = []
    for n_streams in (1, 2, 4, 8):
        chunk = chunk_mb << 20
        dst = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(n_streams)]
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        total = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(2, (2 << 30) // (chunk * n_streams))
        for r in range(reps):
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    off = ((r * n_streams + k) * chunk) % (src.numel() - chunk)
                    dst[k].copy_(src[off:off + chunk], non_blocking=True)
                total += chunk
        torch.cuda.synchronize()
        row.append(total / (time.perf_counter() - t0) / 1e9)
        del dst
    print("  chunk %5d MiB: " % chunk_mb + "  ".join("%d streams %5.1f" % (n, v) for n, v in zip((1, 2, 4, 8), row)), flush=True)

import garbled_snark_verifier_amd as gsv
buf = np.random.default_rng(0).integers(0, 256, (64 << 20,), dtype=np.uint8)
t0 = time.perf_counter()
gsv.cbcmac(buf)
dt = time.perf_counter() - t0
print("host CBC-MAC, one thread: %.3e blocks/s (%.2f GB/s)" % (buf.size / 16 / dt, buf.size / dt / 1e9))
for T in (8, 16, 32, 64, 128):
    th = [threading.Thread(target=gsv.cbcmac, args=(buf,)) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("host CBC-MAC, %3d threads: %.3e blocks/s (%.2f GB/s)" % (T, T * buf.size / 16 / dt, T * buf.size / dt / 1e9), flush=True)
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a", " nproc:", os.cpu_count(), " affinity:", len(os.sched_getaffinity(0)))
