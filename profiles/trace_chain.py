#!/usr/bin/env python3
"""Phase timestamps of mlp_chain_pair_kernel (csrc/mlp_chain.hip, -DCHAIN_TRACE build from profiles/build_chain_variants.sh trace):

    python profiles/trace_chain.py [fwd|bwd]

Workgroup 0, wave 0 (group 0) and wave 4 (group 1), chunks 8..15; cycles (s_memtime) of: the MFMA phase (top -> last product
issued), its waits (vmcnt / lgkmcnt), the barrier behind it, the VALU phase, the barrier behind that one."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from profiles import bench_chain  # noqa: E402

P = ctypes.c_void_p


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
    path = os.path.join(ROOT, 'build_variants', 'libvkas_chaintrace.so')
    M, C = 65536, 384
    bench_chain.run(path, M, C)   # runs fwd then bwd, 11 launches each: the buffer holds the last launch (bwd)
    lib = ctypes.CDLL(path)
    if which == 'fwd':  # one more forward launch so that the buffer holds a forward
        os.environ['CHAIN_TRACE_ONLY'] = 'fwd'
        bench_chain.run(path, M, C, only='fwd')
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    lib.vkas_chain_trace_read.argtypes = [P, ctypes.c_size_t]
    rc = lib.vkas_chain_trace_read(buf, ctypes.sizeof(buf))
    assert rc == 0, rc
    for g in range(2):
        print(f'group {g} (wave {4 * g}), {which}: chunk  mfma  waits  barrier  valu (to h staged / gelu / rest)  barrier   total')
        for c in range(8):
            t = [buf[(g * 8 + c) * 8 + s] for s in range(8)]
            print('   %5d %6d %6d %7d %6d (%5d /%5d /%5d) %7d   %6d' % (8 + c, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3],
                                                                      t[6] - t[3], t[7] - t[6], t[4] - t[7], t[5] - t[4], t[5] - t[0]))


if __name__ == '__main__':
    main()
