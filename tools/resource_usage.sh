# Per-kernel register / scratch / LDS usage of libptgpu's device code (compile only, no GPU needed).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fPIC $EXTRA --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
  -o /tmp/ru/ptgpu.o path-tracer_amd/csrc/ptgpu.hip 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: [^ ]* *//' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/  */ /g'
