"""Build libqagnn_hip.so (gfx950) in-tree with hipcc.  `python -m qagnn_amd.build [--force]`.

The library is rebuilt whenever the SHA-256 of its sources differs from the one recorded next to it at build time
(`libqagnn_hip.so.srchash`): a binary that travelled with the tree (git-ignored, but shipped to the GPU box) is reused only when it
was built from exactly these sources, whatever the file times say."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libqagnn_hip.so')
STAMP = LIB + '.srchash'
SOURCES = ['graph_prep.hip', 'gemm.hip', 'elementwise.hip', 'edge_attn.hip', 'pool.hip', 'hop.hip', 'optim.hip', 'gemm_split.hip', 'gemm_nn2.hip', 'timing.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared']


def source_hash():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    paths = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'common.h'), os.path.join(HERE, '..', 'include', 'qagnn_hip.h')]
    for p in paths:
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read() + b'\0')
    return h.hexdigest()


def up_to_date():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == source_hash()


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library; returns its path."""
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc] + FLAGS + ['-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    if os.path.exists(STAMP):
        os.remove(STAMP)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as f:
        f.write(source_hash() + '\n')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
