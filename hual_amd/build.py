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
FILE_FLAGS = {'convblock.hip': ['-mllvm', '-amdgpu-sched-strategy=max-ilp'],
              # heads.hip without packed fp32 math.  The matching head's float4 arithmetic (f.y * w.xy, p1 * e.xy) was compiled to
              # v_pk_mul/fma_f32 with op_sel:[0,1,..] - the LOW lane of the pair reading the HIGH register of a source pair - the
              # only such instructions in the library, and exactly those instructions lost their low result in lanes 48-63 of a
              # wave whenever another wave of the CU executed matrix instructions - a second queue running MFMA kernels (12 % of forwards with a
              # second stream in the process; scripts/exp/opsel_repro.hip reproduces it without the library: 1500 of 1500 launches beside a v_mfma loop,
              # 0.3-3 % with a second process; never alone, never with this flag: profiles/r6_packed_fp32_opsel.txt).  The
              # kernels of this file are latency bound: no launch got slower (same-box A/B +-0.3 us).  _check_isa() refuses the
              # instruction form in every file.
              'heads.hip': ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']}
OBJDUMP = os.environ.get('HUAL_OBJDUMP', '/opt/rocm/lib/llvm/bin/llvm-objdump')


def _check_isa(obj):
    """refuse a device object holding a packed-fp32 instruction whose op_sel makes the low lane read the high source register (see
    FILE_FLAGS['heads.hip']): on gfx950 its low result is wrong in lanes 48-63 whenever another wave of the SIMD executes matrix instructions -
    a neighbour from another queue or a wave of the same workgroup (scripts/exp/opsel_repro.hip)"""
    import re
    import shutil
    import tempfile
    if os.environ.get('HUAL_BUILD_NO_ISA_CHECK') == '1':      # experiment builds that WANT the refused form (scripts/exp/race_corunner.py)
        return
    if not os.path.exists(OBJDUMP):      # (the ROCm image holds it; without the tool the build goes on and says so)
        sys.stderr.write('[hual build] %s not found: the packed-fp32 op_sel check of %s was skipped\n' % (OBJDUMP, os.path.basename(obj)))
        return
    tmp = tempfile.mkdtemp(prefix='hual_isa_')
    try:
        o = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.run([OBJDUMP, '--offloading', o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        cos = [f for f in glob.glob(o + '.*') if 'amdgcn' in f]
        if not cos:      # a .hip file without kernels (host code only) holds no device bundle
            src = os.path.join(CSRC, os.path.basename(obj)[:-2])
            if '__global__' in open(src).read():
                raise RuntimeError('no gfx950 code object found in %s' % obj)
            return
        bad = []
        for co in cos:
            dis = subprocess.run([OBJDUMP, '-d', co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
            bad += [l.strip() for l in dis.splitlines() if re.search(r'v_pk_((fma|mul|add)_f32|mov_b32) .*op_sel:\[', l)]      # (v_pk_mov_b32: the same operand select; none in the library)
        if bad:
            raise RuntimeError('%s: %d packed-fp32 instructions with op_sel (low lane reads the high register), e.g. "%s": compile the file '
                               'with -packed-fp32-ops (FILE_FLAGS) or restate the arithmetic' % (os.path.basename(obj), len(bad), bad[0]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _stamp(src):
    h = hashlib.sha1()
    for f in [src] + sorted(glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc'))) + [os.path.join(HERE, '..', 'include', 'hual_seqpan.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS + FILE_FLAGS.get(os.path.basename(src), [])).encode())
    return h.hexdigest()


def build(verbose=False, force=False, out=None, defines=(), flags=(), file_flags=None):
    """out / defines: a VARIANT of the current tree (scripts/exp A/B builds) - the same global and per-file flags plus -D switches,
    objects in a directory of its own next to `out`, the in-tree library untouched"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    lib_path, obj_dir, extra = LIB, OBJ, []
    if out is not None:
        lib_path = os.path.abspath(out)
        obj_dir = lib_path + '.obj'
        extra = ['-D' + d for d in defines] + list(flags)
    os.makedirs(obj_dir, exist_ok=True)
    objs, rebuilt, procs = [], False, []
    for src in _sources():
        base = os.path.basename(src)
        obj = os.path.join(obj_dir, base + '.o')
        stamp_file = obj + '.stamp'
        ff = list((file_flags or {}).get(base, []))
        stamp = _stamp(src) + ' '.join(extra + ff)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
            continue
        cmd = [hipcc] + FLAGS + extra + FILE_FLAGS.get(base, []) + ff + (['-x', 'hip'] if src.endswith('.hip') else []) + ['-c', src, '-o', obj]
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
        if base.endswith('.hip'):
            _check_isa(os.path.join(obj_dir, base + '.o'))
        with open(stamp_file, 'w') as fh:
            fh.write(stamp)
    if rebuilt or not os.path.exists(lib_path):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == '__main__':
    # python -m hual_amd.build [--force] [--out build_exp/x.so --define A --define B=1 --flags "-mllvm -x" --file-flags "heads.hip=-mllvm -y" ...]
    args = sys.argv[1:]
    out = args[args.index('--out') + 1] if '--out' in args else None
    defs = [args[i + 1] for i, a in enumerate(args) if a == '--define']
    flags = [f for i, a in enumerate(args) if a == '--flags' for f in args[i + 1].split()]
    ffl = {args[i + 1].split('=', 1)[0]: args[i + 1].split('=', 1)[1].split() for i, a in enumerate(args) if a == '--file-flags'}
    print(build(verbose=out is None, force='--force' in args, out=out, defines=defs, flags=flags, file_flags=ffl))
