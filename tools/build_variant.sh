#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]: build tools/ubench/libzoic_<name>.so from the tree as it stands
# (select it with ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_<name>.so; tools/kbench.py, bench.py and the tests honour that)
N=$1; shift
python - "$N" "$@" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from zoic_amd import build as B
out = os.path.join(os.getcwd(), "tools", "ubench", "libzoic_%s.so" % sys.argv[1])
B.build(force=True, extra_flags=sys.argv[2:], out=out, objdir=os.path.join(os.getcwd(), "tools", "ubench", "obj_" + sys.argv[1]))
print(out)
PY
