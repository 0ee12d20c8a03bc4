# A/B of CDET_HALO_WG3 (three workgroups per CU for the 96-cout patch form of conv_halo_kernel) + the two-stage-ring shapes after the counted-wait fix
python -m pytest tests/test_gpu_conv_tiled.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do
  echo "CDET_HALO_WG3=$v"
  CDET_HALO_WG3=$v python tools/conv_tiled_bench.py --shape 160,160,80,80,3 --shape 80,80,80,80,3 --shape 80,80,160,160,3 --shape 80,80,320,320,3 2>&1 | grep -v amdgpu.ids
done
