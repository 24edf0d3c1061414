"""Build libdep_rnn.so (hipcc, gfx950 only) in-tree next to this file.

    python icassp2022-depression_amd/build_ext.py [--force]

The shared library is plain C-ABI (include/dep_rnn.h): no torch, no pybind.  hipcc cross-compiles
without a GPU, so this also runs in the CPU-only build container (__graft_entry__.build()).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libdep_rnn.so')
SOURCES = ['api.hip', 'gemm.hip', 'gemm_bf16x3.hip', 'rnn_sweep.hip', 'rnn_cluster.hip', 'rnn_cluster_bwd.hip', 'rnn_cluster16.hip', 'rnn_cluster_lstm.hip', 'rnn_fused2.hip', 'rnn_fused2_bwd.hip', 'elementwise.hip', 'head.hip', 'attention.hip', 'frontend.hip', 'comm.hip']
HEADERS = ['dep_common.h', 'rnn_cluster_common.h', os.path.join('..', '..', 'include', 'dep_rnn.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result',
         '-Wno-pass-failed']


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + '.stamp'
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    with open(stamp, 'w') as f:
        f.write(dig)
    if verbose:
        print(f'built {LIB} ({os.path.getsize(LIB) / 1e6:.2f} MB)')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
