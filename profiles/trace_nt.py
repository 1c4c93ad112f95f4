#!/usr/bin/env python3
"""Where a tile of gemm_nt_mfma_kernel spends its time: runs one NT GEMM shape on the -DVKAS_TRACE build
(profiles/build_trace.sh; VKAS_LIB_PATH=build_variants/libvkas_trace.so) and prints, over all workgroups, the median
duration of the phases between the kernel's timestamps and how the workgroups follow each other on a CU.
usage: trace_nt.py M N K mode"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vkit_ocr_model_adaptive_scaling_amd import _lib  # noqa: E402

lib = _lib.lib
M, N, K, mode = (int(a) for a in sys.argv[1:5])
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn((M, K), generator=g, device='cuda').bfloat16()
w = (torch.randn((N, K), generator=g, device='cuda') * 0.05).bfloat16()
out, out2 = (torch.empty((M, N), device='cuda', dtype=torch.bfloat16) for _ in range(2))
aux = torch.randn((M, N), generator=g, device='cuda').bfloat16()
bias, cs = torch.zeros((N,), device='cuda'), torch.ones((N,), device='cuda')
geom = _lib.ConvGeom(1, M // 256, 256, M // 256, 256, K, K, 1, 1, 1, 0)
epi = _lib.Epilogue()
epi.mode, epi.out, epi.ldo, epi.bias = mode, out.data_ptr(), N, bias.data_ptr()
if mode == 1:
    epi.out2, epi.ldo2 = out2.data_ptr(), N
if mode in (2, 3, 4):
    epi.aux, epi.ldaux = aux.data_ptr(), N
if mode == 2:
    epi.colscale, epi.rows_per_image = cs.data_ptr(), 65536


def run():
    _lib.check(lib.vkas_conv_gemm_fwd(ctypes.c_void_p(x.data_ptr()), ctypes.byref(geom), ctypes.c_void_p(w.data_ptr()), N,
                                      ctypes.byref(epi), _lib.BF16, st), 'conv_gemm_fwd')


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
run()
e.record()
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ['VKAS_LIB_PATH'])
buf = np.zeros(65536 * 8, dtype=np.uint64)
assert raw.vkas_trace_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
t = buf.reshape(-1, 8)
nb = int((t[:, 0] != 0).sum())
t = t[:nb].astype(np.int64)
if t[:, 5].max() > 0 and t[:, 5].max() < 65536:  # a persistent kernel recorded its tile count (experiments/gemm_nt_stream_kernel.patch)
    print(f'  (streaming kernel: per workgroup, summed over its {int(np.median(t[:, 5]))} tiles: K loop = all K loops, epilogue = all epilogues)')
print(f'M={M} N={N} K={K} mode={mode}: {nb} workgroups, launch {s.elapsed_time(e) * 1e3:.1f} us')
xcc = t[:, 7] & 0xf
# the counters of different XCDs are not aligned: calibrate the tick on the XCD with the longest first start -> last end
# (event time of the launch ~ that span) and never compare timestamps across XCDs
span = max(float(t[xcc == c][:, 4].max() - t[xcc == c][:, 0].min()) for c in np.unique(xcc))
tick = 1.0 / 2100.0  # s_memtime counts shader clocks (~2.1 GHz under load); spans across CUs are not comparable

names = ['prologue (first tiles landed)', 'K loop', 'epilogue (stores issued)', 'stores acknowledged']
for i, n in enumerate(names):
    d = (t[:, i + 1] - t[:, i]) * tick
    print(f'  {n:32s} median {np.median(d):7.2f} us ({np.median(d) / tick:8.0f} cycles)   p10 {np.percentile(d, 10):7.2f}   p90 {np.percentile(d, 90):7.2f}')
life = (t[:, 4] - t[:, 0]) * tick
print(f'  workgroup lifetime               median {np.median(life):7.2f} us')
# successive workgroups of one CU: (xcc, se, sh, cu) from XCC_ID / HW_ID
hw = t[:, 6]
cu = xcc * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xf)
gaps, per_cu = [], []
for c in np.unique(cu):
    rows = t[cu == c]
    rows = rows[np.argsort(rows[:, 0])]
    per_cu.append(len(rows))
    gaps.extend(((rows[1:, 0] - rows[:-1, 4]) * tick).tolist())
if not gaps:
    gaps = [0.0]
print(f'  {len(np.unique(cu))} CUs, workgroups per CU {min(per_cu)}..{max(per_cu)}; gap between a workgroup\'s end and the next start '
      f'on its CU: median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f} us')
