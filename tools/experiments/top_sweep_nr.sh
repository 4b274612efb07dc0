# where the multifrontal top starts and how large its fronts may get (items per level / parallel chains per level / soft front cap):
# factorisation time.  B=64 tools/top_sweep_nr.sh  or  B=512 ...
B=${B:-512}
if [ "$B" -ge 256 ]; then CFGS="280,4,47 280,6,47 280,8,47 280,0,47 384,0,47 384,8,47 512,0,47 280,4,40 384,8,32 384,0,24 600,0,47"
else CFGS="384,0,24 384,0,28 384,0,32 384,0,40 384,0,47 280,0,32 512,0,32 600,0,40 384,4,32 280,4,47"; fi
for cfg in $CFGS; do IFS=, read i c f <<< "$cfg"
  echo "items $i chains $c front $f: $(JG_TOP_ITEMS=$i JG_TOP_CHAINS=$c JG_TOP_FRONT=$f python tools/time_kernels.py $B ${CASE:-case_ACTIVSg10k} 10 2>&1 | tail -1)"
done
