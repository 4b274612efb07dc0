JG_TOP_PROFILE=1 python tools/time_se.py ${1:-512} 2>&1 | grep "top profile\|rows " | cut -c1-230
