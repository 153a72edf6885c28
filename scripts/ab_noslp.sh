# every kernel file built with -fno-slp-vectorize against the in-tree build (phmm's gain: profiles/r06k_phmm_ab_libs.txt)
for k in abea chain fmi poa bsw; do
  bash scripts/ab_libs.sh r06l_${k}_noslp $k 2 intree build_tmp/libgbx_${k}_noslp.so > /dev/null 2>&1
  cat gpurun_out/r06l_${k}_noslp_ab_libs.txt | cut -c1-220
done
