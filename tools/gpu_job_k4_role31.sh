cd "$GRAFT_REPO_ROOT"; R=$PWD
for V in k4r0 k4r2 k4r1; do echo $V; WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so python tools/k4_planes_bits.py 2>&1 | grep -v amdgpu.ids; done
bash tools/gpu_job_kvar.sh k4xsplit "xty_f16x3_kernel<256, true" k4r0 k4r2 k4r1
bash tools/gpu_job_kvar.sh k4bits "xty_f16x3_kernel<256, true" k4r0 k4r1
bash tools/gpu_job_kvar.sh k4 "xty_f16x3_kernel<256, true" k4r0 k4r1
