"""Build a variant of libsnmf_hip.so beside the product library (never over it), for experiments and the
diagnostic -DSNMF_PROF build:  python scripts/build_variant.py NAME [hipcc flags ...]
    -> scripts/prof_build/libsnmf_NAME.so (objects under build/obj_NAME/), selected with SNMF_LIB_PATH.
`prod` as NAME rebuilds the product library itself (se_snmf_nat_amd/libsnmf_hip.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se_snmf_nat_amd import _lib  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
if name == "prod":
    print(_lib.build(verbose=False, extra_flags=flags))
else:
    out_dir = os.path.join(ROOT, "scripts", "prof_build")
    os.makedirs(out_dir, exist_ok=True)
    print(_lib.build(verbose=False, extra_flags=flags, lib_path=os.path.join(out_dir, f"libsnmf_{name}.so"),
                     obj_dir=os.path.join(ROOT, "build", f"obj_{name}")))
