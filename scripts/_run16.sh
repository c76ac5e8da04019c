export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for v in w6 w7 bufl w6 w7 bufl; do echo "== $v"; ACX_LIB=build_ab/libacx_$v.so python scripts/ab_narrow.py 9 2>/dev/null; done
