for pw in 0 1; do for H in 0 3 0 3; do
  if [ $H != 0 ]; then export JG_LIB=$PWD/build/libjg_probe_top$H.so; else unset JG_LIB; fi
  echo "JG_TOP_PW=$pw JG_PROBE_TOP=$H $(JG_TOP_PW=$pw python tools/time_kernels.py 512 case_ACTIVSg10k 20 2>&1 | tail -1) | $(JG_TOP_PW=$pw python tools/time_kernels.py 1 case_ACTIVSg10k 20 2>&1 | tail -1)"
done; done
unset JG_LIB
for H in 0 3; do if [ $H != 0 ]; then export JG_LIB=$PWD/build/libjg_probe_top$H.so; else unset JG_LIB; fi; echo "SE JG_PROBE_TOP=$H $(python tools/time_se.py 512 2>&1 | grep 'rows ' | tail -1)"; done
