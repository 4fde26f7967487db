#!/bin/bash
# Round 6 experiment: issue priorities in the policy kernel (two 4-wave blocks per CU: two waves per SIMD, one of each block).
#   p1: the younger block of a CU's pair at s_setprio 1 (MI355X_MICROARCH.md "static priority for the younger half")
#   p2: priority falls layer by layer (the block that is behind goes first)      p3: priority rises layer by layer
# Variant libraries are built here from patched copies of eb_policy.hip.
TAG=${1:-r6mlp}; OUT=gpurun_out/$TAG; mkdir -p $OUT /tmp/mlp
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; echo "build rc=$?"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-kernarg-preload-count=14"
S=env_build_amd/csrc
for f in eb_capi eb_kernels eb_rollout eb_env_kernels eb_env_step eb_env_step_t1 eb_env_step_t2; do /opt/rocm/bin/hipcc $F -c $S/$f.hip -o /tmp/mlp/$f.o 2>> $OUT/build.log & done
for v in 1 2 3; do
  mkdir -p /tmp/mlp/src$v; cp $S/*.h $S/eb_policy.hip /tmp/mlp/src$v/
  python - <<PY
p='/tmp/mlp/src$v/eb_policy.hip'
s=open(p).read()
v=$v
if v == 1:
    old='''    // ---- hidden layers ----'''
    new='''    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(1);
    // ---- hidden layers ----'''
    assert old in s; s=s.replace(old,new,1)
else:
    old='''        const MlpLayer& ly = A.hid[L];
        f32x16 acc[RT][CT];'''
    new='''        const MlpLayer& ly = A.hid[L];
        if (L == 0) __builtin_amdgcn_s_setprio(%d); else __builtin_amdgcn_s_setprio(%d);
        f32x16 acc[RT][CT];''' % ((2, 1) if v == 2 else (0, 1))
    assert old in s; s=s.replace(old,new,1)
    old='''    // ---- output layer: 16 x 16 x 4 tiles'''
    new='''    __builtin_amdgcn_s_setprio(%d);
    // ---- output layer: 16 x 16 x 4 tiles''' % (0 if v == 2 else 2)
    assert old in s; s=s.replace(old,new,1)
open(p,'w').write(s)
PY
  /opt/rocm/bin/hipcc $F -c /tmp/mlp/src$v/eb_policy.hip -o /tmp/mlp/policy_p$v.o 2>> $OUT/build.log &
done
wait
for v in 1 2 3; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/mlp/eb_*.o /tmp/mlp/policy_p$v.o -o /tmp/mlp/libp$v.so 2>> $OUT/build.log; done
ls -la /tmp/mlp/*.so; tail -3 $OUT/build.log
{
for rep in 1 2 3; do
  for x in base p1 p2 p3; do
    lib=""; [ $x != base ] && lib="--lib /tmp/mlp/lib$x.so"
    echo "rep $rep x=$x: $(timeout 300 python scripts/time_policy.py $lib 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-600)"
  done
done
} 2>&1 | tee $OUT/ab.txt
