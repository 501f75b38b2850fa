"""CPU: the build's instruction guard (hual_amd/build.py _check_isa).  Packed-fp32 instructions whose op_sel makes the low lane read the
high register of a source pair lost results on gfx950 when two queues shared the GPU (profiles/r6_packed_fp32_opsel.txt): the build
refuses them, and every object of the in-tree library passes."""
import glob
import os
import subprocess

import pytest

from hual_amd import build

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
SRC = """
#include <hip/hip_runtime.h>
__global__ void k(const float4* f, const float4* w, float4* o) {      // the matching head's logits: a.y * (w1.x, w1.y) reads the HIGH
  const float4 a = f[threadIdx.x];                                     // register of the (a.x, a.y) pair for the low lane
  const float4 w0 = w[4 * threadIdx.x], w1 = w[4 * threadIdx.x + 1], w2 = w[4 * threadIdx.x + 2], w3 = w[4 * threadIdx.x + 3];
  o[threadIdx.x] = make_float4(a.x * w0.x + a.y * w1.x + a.z * w2.x + a.w * w3.x, a.x * w0.y + a.y * w1.y + a.z * w2.y + a.w * w3.y,
                               a.x * w0.z + a.y * w1.z + a.z * w2.z + a.w * w3.z, a.x * w0.w + a.y * w1.w + a.z * w2.w + a.w * w3.w);
}
"""


def test_objects_of_the_library_hold_no_refused_instruction():
    objs = sorted(glob.glob(os.path.join(build.OBJ, '*.hip.o')))
    if not objs:
        build.build()
        objs = sorted(glob.glob(os.path.join(build.OBJ, '*.hip.o')))
    assert len(objs) >= 10
    for o in objs:
        build._check_isa(o)


def test_guard_refuses_the_instruction_form(tmp_path):
    src = tmp_path / 'probe.hip'
    src.write_text(SRC)
    flags = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-x', 'hip', '-c']
    obj = tmp_path / 'probe.hip.o'
    subprocess.check_call([HIPCC] + flags + [str(src), '-o', str(obj)], stderr=subprocess.DEVNULL)
    asm = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip', '-S', '--cuda-device-only', str(src), '-o', '-'],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    if 'op_sel:[' not in asm:
        pytest.skip('this compiler does not pair the probe into the op_sel form')
    old = build.CSRC
    try:
        build.CSRC = str(tmp_path)          # where the guard looks for the source of an object without device code
        with pytest.raises(RuntimeError, match='op_sel'):
            build._check_isa(str(obj))
        # the same source without packed fp32 math (what FILE_FLAGS puts on heads.hip) passes
        obj2 = tmp_path / 'probe2.hip.o'
        subprocess.check_call([HIPCC] + flags + build.FILE_FLAGS['heads.hip'] + [str(src), '-o', str(obj2)], stderr=subprocess.DEVNULL)
        (tmp_path / 'probe2.hip').write_text(SRC)
        build._check_isa(str(obj2))
    finally:
        build.CSRC = old
