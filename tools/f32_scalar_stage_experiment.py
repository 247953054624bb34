"""VERDICT r04 #2b, measurement only: builds a copy of the library in which the scalar stage of the ICP half-iteration
prologue (6x6 solve, both SE(3) exponentials, the gradLM update: gs_icp_math.h gs_solve_spd6_wave / gs_se3_exp_wave /
icp_update_math_wave and the exp / log / sincos polynomials they call) runs in float32 instead of float64; the float64
row sums stay.  NOT a product path and NOT bit-compatible with the oracle: it exists to measure what the float64 chain
costs per launch (run next to the product build through GRADSLAM_HIP_LIB; results in
profiles/r05_f32_scalar_stage_experiment.txt).

    python tools/f32_scalar_stage_experiment.py <scratch dir>      -> <scratch dir>/gradslam_amd/csrc/libgradslam_hip.so
"""
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = sys.argv[1]
shutil.rmtree(dst, ignore_errors=True)
os.makedirs(dst)
for d in ("gradslam_amd", "include"):
    shutil.copytree(os.path.join(ROOT, d), os.path.join(dst, d), ignore=shutil.ignore_patterns("*.o", "*.so", "__pycache__", ".build_flags"))
p = os.path.join(dst, "gradslam_amd", "csrc", "gs_icp_math.h")
s = open(p).read()
lit = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")


def floats(body):   # every floating literal becomes a float literal
    return lit.sub(lambda m: m.group(1) + "f", body)


def span(src, sig):
    i = src.index("GS_DEV " + sig)
    return i, src.index("\n}\n", i) + 3


twins = ""
for sig in ("void gs_sincos_fast(", "double gs_exp_fast(", "double gs_log_fast("):
    i, j = span(s, sig)
    b = s[i:j]
    for n in ("gs_sincos_fast", "gs_exp_fast", "gs_log_fast"):
        b = b.replace(n, n + "_f32")
    b = re.sub(r"\bdouble\b", "float", b)
    for a, c in (("sin(x)", "sinf(x)"), ("cos(x)", "cosf(x)"), ("exp(x)", "expf(x)"), ("log(y)", "logf(y)"), ("fabs(", "fabsf("),
                 ("rint(", "rintf("), ("ldexp(", "ldexpf("), ("frexp(", "frexpf("), ("1e-300", "1e-30"), ("1e300", "1e30"), ("700.0", "80.0")):
        b = b.replace(a, c)
    b = re.sub(r"1\.0 / (\d{11,}\.0)", lambda m: "(float)(1.0 / %s)" % m.group(1), b)   # (factorials beyond float range)
    parts = re.split(r"(\(float\)\(1\.0 / \d+\.0\))", b)
    twins += "".join(x if x.startswith("(float)(1.0 /") else floats(x) for x in parts) + "\n"
k = s.index("// geometry/se3utils.py:77-115 in double, rounded once")
s = s[:k] + twins + s[k:]
for sig in ("void gs_se3_exp_wave(", "void gs_solve_spd6_wave(", "void icp_update_math_wave("):
    i, j = span(s, sig)
    b = re.sub(r"\bdouble\b", "float", s[i:j])
    for n in ("gs_sincos_fast(", "gs_exp_fast(", "gs_log_fast("):
        b = b.replace(n, n[:-1] + "_f32(")
    s = s[:i] + floats(b.replace("sqrt(w[0]", "sqrtf(w[0]")) + s[j:]
s = s.replace("GS_DEV void gs_solve_spd6_wave(const float* S,", "GS_DEV void gs_solve_spd6_wave(const double* S,")
open(p, "w").write(s)
sys.path.insert(0, dst)
from gradslam_amd.csrc import build   # noqa: E402  (the copy's build script)
print(build.build(force=True))
