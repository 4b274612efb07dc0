for b in ${1:-512 64}; do echo "== b=$b"; JG_TOP_PROFILE=1 python tools/time_kernels.py $b ${2:-case_ACTIVSg10k} 5 2>&1 | grep "top profile" | cut -c18-; done
