#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]: build tools/ubench/libzoic_<name>.so (select with ZOIC_AMD_LIB=...)
N=$1; shift
python - "$N" "$@" <<'PY'
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from zoic_amd import build as B
out = os.path.join("tools", "ubench", "libzoic_%s.so" % sys.argv[1])
subprocess.check_call([B._hipcc()] + B.FLAGS + sys.argv[2:] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", out])
print(out)
PY
