"""Builds hual_amd/libhual_seqpan.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libhual_seqpan.so')
OBJ = os.path.join(HERE, 'csrc', 'build')
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (the kernels' epilogues are VALU work on the accumulators: in the AGPR form
# every accumulator register paid a v_accvgpr_read - one vector instruction per attention score; no kernel needs more than 256 registers)
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function', '-mllvm', '-amdgpu-mfma-vgpr-form']
if os.environ.get('HUAL_STAMPS'):        # debug build: in-kernel phase timestamps of one fused tile kernel (csrc/tilecore.h)
    FLAGS.append('-DHUAL_STAMPS=' + os.environ['HUAL_STAMPS'])
    if os.environ.get('HUAL_STAMPS_FIRST'):
        FLAGS.append('-DHUAL_STAMPS_FIRST')
if os.environ.get('HUAL_EXP_DEFS'):      # experiment builds: extra -D switches (scripts/exp)
    FLAGS.extend('-D' + d for d in os.environ['HUAL_EXP_DEFS'].split(','))


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


# per-file additions.  convblock.hip: the machine scheduler's max-ILP strategy (conv_block_bwd 3 x 39.5 -> 3 x 38.8 us, conv_block_fwd -0.3 us
# per launch in a same-box A/B of the whole library built with it; the other files gain nothing or lose: mproj +1..2 us, dw +0.7)
FILE_FLAGS = {'convblock.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp']}


def _stamp(src):
    h = hashlib.sha1()
    for f in [src] + sorted(glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))) + [os.path.join(HERE, '..', 'include', 'hual_seqpan.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS + FILE_FLAGS.get(os.path.basename(src), [])).encode())
    return h.hexdigest()


def build(verbose=False, force=False, out=None, defines=()):
    """out / defines: a VARIANT of the current tree (scripts/exp A/B builds) - the same global and per-file flags plus -D switches,
    objects in a directory of its own next to `out`, the in-tree library untouched"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    lib_path, obj_dir, extra = LIB, OBJ, []
    if out is not None:
        lib_path = os.path.abspath(out)
        obj_dir = lib_path + '.obj'
        extra = ['-D' + d for d in defines]
    os.makedirs(obj_dir, exist_ok=True)
    objs, rebuilt, procs = [], False, []
    for src in _sources():
        base = os.path.basename(src)
        obj = os.path.join(obj_dir, base + '.o')
        stamp_file = obj + '.stamp'
        stamp = _stamp(src) + ' '.join(extra)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
            continue
        cmd = [hipcc] + FLAGS + extra + FILE_FLAGS.get(base, []) + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), stamp_file, stamp, base))
        rebuilt = True
    for pr, stamp_file, stamp, base in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % base)
        if verbose and out:
            print(out.decode())
        with open(stamp_file, 'w') as fh:
            fh.write(stamp)
    if rebuilt or not os.path.exists(lib_path):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == '__main__':
    # python -m hual_amd.build [--force] [--out build_exp/x.so --define A --define B=1 ...]
    args = sys.argv[1:]
    out = args[args.index('--out') + 1] if '--out' in args else None
    defs = [args[i + 1] for i, a in enumerate(args) if a == '--define']
    print(build(verbose=out is None, force='--force' in args, out=out, defines=defs))
