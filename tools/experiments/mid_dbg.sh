cd "$(dirname "$0")/.."
run() { # case batch envs...
  case=$1; b=$2; shift 2
  python tools/mid_debug.py $case $b /tmp/ref.npy > /dev/null
  echo "== $case $b $*"
  env "$@" python tools/mid_debug.py $case $b /tmp/m.npy && python tools/mid_debug.py --compare /tmp/ref.npy /tmp/m.npy
}
run case14 64 JG_TOP_LEVEL=1 JG_MID_STRUCT=1 JG_MID_MMIN=1 JG_MID_NOGROUP=3
run case118 64 JG_TOP_LEVEL=1 JG_MID_STRUCT=1 JG_MID_MMIN=1 JG_MID_NOGROUP=3
run case1354pegase 64 JG_MID_STRUCT=3 JG_MID_MMIN=4 JG_MID_NOGROUP=3
