export CDET_LIB_PATH=tools/debug/_build/libcdet_prof.so
for v in 0 1; do
  echo "== CDET_HALO_WG3=$v"
  CDET_HALO_WG3=$v python tools/halo_timeline.py --custom 160,160,80,80,3 --mode silu 2>&1 | grep -v amdgpu.ids
  CDET_HALO_WG3=$v python tools/halo_timeline.py --custom 160,160,80,80,3 --mode raw 2>&1 | grep -v amdgpu.ids
done
