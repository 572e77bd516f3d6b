mkdir -p gpurun_out/r05
for lib in "" vqvae_amd/build/variants/libvqvae_pipe_noacq.so vqvae_amd/build/variants/libvqvae_pipe_nofence.so; do
  echo "== lib: ${lib:-default}"
  VQVAE_HIP_LIB_OVERRIDE=$lib timeout 200 python tools/experiments/r05_pipe_ab.py 4096 1024 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05/pipe_ab.txt
