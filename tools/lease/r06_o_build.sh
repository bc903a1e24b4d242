#!/bin/bash
# builds libecfft_hip with extra backend flags into ecfft_amd/variants/cf_NAME.so
cd "$(dirname "$0")/../.."; mkdir -p ecfft_amd/variants
SRC=ecfft_amd/csrc/ecfft_capi.hip
build() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed "$@" -o ecfft_amd/variants/cf_$name.so $SRC > /tmp/cf_$name.log 2>&1; echo "$name rc=$?"; }
build ilp -mllvm -amdgpu-sched-strategy=max-ilp &
build memcl -mllvm -amdgpu-sched-strategy=max-memory-clause &
build iterilp -mllvm -amdgpu-sched-strategy=iterative-ilp &
build bias0 -mllvm -amdgpu-schedule-metric-bias=0 &
wait
build nopost -mllvm -enable-post-misched=false &
build trackers -mllvm -amdgpu-use-amdgpu-trackers &
build relaxed -mllvm -amdgpu-schedule-relaxed-occupancy &
build bias100 -mllvm -amdgpu-schedule-metric-bias=100 &
wait
ls -la ecfft_amd/variants/
