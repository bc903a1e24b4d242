#!/bin/bash
# round 6, session 4: tools/ubench/xvec_stage.hip — one butterfly stage on the matrix cores with the constant shared across 32 vectors, against the
# shipped table multiply on the same work (a measurement under the estimate)
O=gpurun_out/r06s; rm -rf $O; mkdir -p $O
for mb in 2 3 4; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMINB=$mb -Wno-pass-failed -o /tmp/xvec_mb$mb tools/ubench/xvec_stage.hip 2>/dev/null || echo "build mb$mb failed"; done
{
for mb in 2 3 4; do for ng in 16 256 2048; do echo "## workgroups per CU (launch bound) $mb, constant groups $ng"; timeout 300 /tmp/xvec_mb$mb 8192 50 $ng; done; done
} > $O/ubench_xvec_stage.txt 2>&1
cat $O/ubench_xvec_stage.txt
