"""Build libqagnn_hip.so (gfx950) in-tree with hipcc.  `python -m qagnn_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libqagnn_hip.so')
SOURCES = ['graph_prep.hip', 'gemm.hip', 'elementwise.hip', 'edge_attn.hip', 'pool.hip', 'hop.hip', 'optim.hip', 'gemm_split.hip']


def _newest_source_mtime():
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'common.h'),
                                                         os.path.join(HERE, '..', 'include', 'qagnn_hip.h')]
    return max(os.path.getmtime(p) for p in paths)


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source_mtime():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-o', LIB] + \
          [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
