"""The lattice build's radix sort (csrc/sort.hip) at the bench's sizes: 23.7 M (key, value) pairs over 55 key bits in 35 segments
(the bilateral entries of a 35-image batch), 3.6 M pairs over 40 bits (the renumbering sort), and the int32 prefix sum."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pnp-ovss_amd"))
import torch
from pnp_ovss import hip

lib = hip.load_library()


def sort_case(n, bits, nseg):
    g = torch.Generator(device="cuda").manual_seed(0)
    keys = torch.randint(0, 2**62, (n,), device="cuda", generator=g, dtype=torch.int64) & ((1 << bits) - 1)
    vals = torch.arange(n, device="cuda", dtype=torch.int32)
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    off = (ctypes.c_size_t * (nseg + 1))(*[n * i // nseg for i in range(nseg + 1)])
    best = 1e9
    for rep in range(4):
        k, v = keys.clone(), vals.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert lib.pnp_op_sort_pairs(k.data_ptr(), ko.data_ptr(), v.data_ptr(), vo.data_ptr(), n, 0, bits, off, nseg, None) == 0
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    passes = (bits + 7) // 8
    print(f"sort {n / 1e6:5.1f} M pairs, {bits} bits ({passes} passes), {nseg} segments: {best * 1e3:6.2f} ms incl. scratch allocation "
          f"= {best / passes * 1e6:6.1f} us per pass, {36.0 * n * passes / best / 1e12:.2f} TB/s of the 36 B per item and pass", flush=True)


def scan_case(n):
    x = torch.ones(n, device="cuda", dtype=torch.int32)
    y = torch.empty_like(x)
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert lib.pnp_op_scan_i32(x.data_ptr(), y.data_ptr(), n, 1, None) == 0
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert int(y[-1]) == n
    print(f"scan {n / 1e6:5.1f} M int32: {best * 1e3:6.2f} ms incl. scratch allocation = {12.0 * n / best / 1e12:.2f} TB/s of 12 B per item", flush=True)


sort_case(23_708_160, 55, 35)
sort_case(23_708_160, 55, 1)
sort_case(3_560_000, 40, 35)
scan_case(23_708_160)
