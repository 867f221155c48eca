"""Builds libfastnerf.so (HIP kernels + C ABI + host quadtree) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the
gpurun snapshot.  No torch headers are involved: the boundary is a plain C ABI
(include/fastnerf.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfastnerf.so')
SOURCES = ['rays.hip', 'composite.hip', 'train.hip', 'mlp_pack.hip', 'mlp_fwd.hip', 'mlp_bwd_dx.hip', 'mlp_bwd_dw.hip', 'mlp_bf16.hip', 'tree.cpp', 'render.cpp', 'comm.cpp']
# -fno-slp-vectorize: with SLP on, hipcc (ROCm 7.2) packs the epilogues' scalar fp32 adds into v_pk_add_f32; on
# gfx950 the split-bf16 forward then produced sporadic wrong sums (bias dropped in lanes 48..63 of one register,
# ~0.1% of points, timing dependent) -- and packed fp32 VALU next to MFMAs is slower anyway.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC',
         '-Wno-unused-result']


def _digest(extra=()):
    """sha256 over the sources the library is made of (+ the flags): what `libfastnerf.so.src` records next to a build."""
    import hashlib
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(HERE, '..', 'include', 'fastnerf.h')]
    for d in deps:
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def _stale(extra=()):
    # by CONTENT, not by mtime: a library that travelled (gpurun snapshot, checkout) is fresh exactly when it was built from these sources
    if not os.path.exists(LIB) or not os.path.exists(LIB + '.src'):
        return True
    with open(LIB + '.src') as f:
        return f.read().strip() != _digest(extra)


def build(force=False, verbose=False):
    extra = os.environ.get('FASTNERF_CFLAGS', '').split()   # tuning experiments
    if not force and not _stale(extra) and not os.environ.get('FASTNERF_VARIANT'):
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if any(f.startswith('-DFASTNERF_ABLATION') for f in extra) and not os.environ.get('FASTNERF_VARIANT'):
        # timing-only builds compute WRONG RESULTS by design (csrc/mlp_common.h): they may only exist as variants/<name>.so
        raise RuntimeError('FASTNERF_ABLATION builds need FASTNERF_VARIANT=<name>: the product library is never an ablation build')
    objs = []
    procs = []
    variant = os.environ.get('FASTNERF_VARIANT')   # tuning builds: variants/<name>.so, loaded with FASTNERF_LIB=...
    bdir = os.path.join(HERE, 'build', variant) if variant else os.path.join(HERE, 'build')
    lib = os.path.join(HERE, 'variants', variant + '.so') if variant else LIB
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src + '.o')
        objs.append(obj)
        cmd = [hipcc] + FLAGS + extra + (['-x', 'hip'] if src.endswith('.cpp') else []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError('link failed')
    with open(lib + '.src', 'w') as f:
        f.write(_digest(extra) + '\n')
    return lib


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
