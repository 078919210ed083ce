# K4's quadrant form: digests of R | gsum (tools/k4_planes_bits.py) and the kernel's durations under rocprofv3 for builds with XTY_ROLE31 = 0 / 1 / 2 / 3
# (tools/build_var.py wc_fast_xty k4r0=-DXTY_ROLE31=0 k4r1=-DXTY_ROLE31=1 ...); usage: gpu_job_k4_role31.sh <tags...>
cd "$GRAFT_REPO_ROOT"; R=$PWD
for V in "$@"; do echo $V; WC_LIB=$R/wc_gan_amd/csrc/build/var/lib_$V.so python tools/k4_planes_bits.py 2>&1 | grep -v amdgpu.ids; done
bash tools/gpu_job_kvar.sh k4xsplit "xty_f16x3_kernel<256, true" "$@"
