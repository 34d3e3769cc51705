"""Build libcrowdstep.so (HIP, gfx950) in-tree with hipcc.  No CPU fallback is ever built.

The rebuild is keyed on CONTENT, not on mtimes: every translation unit is compiled to an object under csrc/.build/ whose
key is sha256(source + the headers it includes + flags); the library carries the combined key (`cs_build_id()`), and
`libcrowdstep.so.manifest.json` beside it records the per-file keys.  `status()` tells whether the library on disk was
built from the sources on disk -- tests/test_abi_cpu.py fails when it was not.
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(CSRC)
ROOT = os.path.dirname(PKG)
LIB_PATH = os.path.join(PKG, "libcrowdstep.so")
MANIFEST_PATH = LIB_PATH + ".manifest.json"
OBJ_DIR = os.path.join(CSRC, ".build")
SOURCES = ["crowdstep.hip", "sfmstep_generic.hip", "sfmstep_leanrt.hip", "sfmstep_lean25.hip", "sfmstep_lean30.hip", "sfmstep_small.hip", "sfmstep_lean50.hip", "sfmstep_robot26.hip", "sfmstep_robotx.hip", "sfmstep_imit.hip", "sfmstep_peragent.hip", "orca.hip", "lookahead.hip", "generate.hip", "laser.hip", "social_momentum.hip", "robot_model.hip",
           "rk45.hip", "rowstep.hip", "gymstep.hip", "bigworld.hip"]
ARCH = "gfx950"
# -fno-slp-vectorize: v_pk_*_f32 has no throughput advantage over two scalar ops on gfx950 (measured,
# tools/valu_microbench.hip) and packing costs ~2 v_mov per partner in the pair loop
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]


# Experiment hook (OFF by default): CROWDSTEP_POSTPASS=1 sends the device code of every translation unit through
# csrc/asm_postpass.py between `hipcc -S --cuda-device-only` and the assembler (hipcc's own pipeline, opened in the middle).
# Round 5 used it for one rule -- back-to-back VOP2 selects on vcc re-encoded as VOP3, 16 -> 4 SIMD cycles each in
# tools/valu_issue_ceiling.hip -- and measured NO change on any kernel of this library (profiles/r5_ab_postpass_and_orca_math.txt),
# so the product build is plain `hipcc -c`.
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def postpass_enabled() -> bool:
    return os.environ.get("CROWDSTEP_POSTPASS") == "1" and all(
        os.path.exists(os.path.join(LLVM_BIN, t)) for t in ("clang", "lld", "clang-offload-bundler"))


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the crowd stepper only exists as a HIP library for gfx950")


def _sources() -> list[str]:
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers() -> list[str]:
    hs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".inc")))
    return hs + [os.path.join(ROOT, "include", "crowdstep.h")]


def _sha(paths: list[str], extra: str = "") -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(extra.encode())
    return h.hexdigest()


def source_keys(extra_flags: list[str] | None = None) -> dict:
    """Per-file content keys and the combined build id of the sources on disk."""
    flags = " ".join(FLAGS + (extra_flags or []))
    if postpass_enabled():
        from . import asm_postpass
        flags += " postpass=" + asm_postpass.VERSION
    hdr = _sha(_headers())
    files = {s: _sha([os.path.join(CSRC, s)], hdr + flags) for s in _sources()}
    build_id = hashlib.sha256(json.dumps(files, sort_keys=True).encode()).hexdigest()[:32]
    return {"files": files, "build_id": build_id, "flags": flags}


def status() -> dict:
    """{'exists', 'fresh', 'build_id' (sources), 'lib_build_id' (manifest beside the library)}"""
    keys = source_keys()
    out = {"exists": os.path.exists(LIB_PATH), "fresh": False, "build_id": keys["build_id"], "lib_build_id": None}
    if out["exists"] and os.path.exists(MANIFEST_PATH):
        try:
            out["lib_build_id"] = json.load(open(MANIFEST_PATH)).get("build_id")
        except Exception:
            pass
    out["fresh"] = out["exists"] and out["lib_build_id"] == keys["build_id"]
    return out


def needs_build() -> bool:
    return not status()["fresh"]


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None, jobs: int | None = None, variant: str | None = None) -> str:
    """Compile what changed and link.  Returns the library path; `build.last_action` says 'compiled' or 'reused'.
    `variant`: an A/B build beside the product library (libcrowdstep_<variant>.so, objects under .build_<variant>/, never claims
    to be the product build; tools load it through CROWDSTEP_LIB)."""
    global last_action
    if variant:
        return _build_variant(variant, verbose, extra_flags, jobs)
    keys = source_keys(extra_flags)
    if extra_flags:   # diagnostic builds (e.g. -DCS_STAMPS) never claim to be the product build
        keys["build_id"] = "diag-" + keys["build_id"][:27]
    if not force and not extra_flags and status()["fresh"]:
        last_action = "reused"
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = hipcc_path()
    todo = []
    objs = []
    for s, key in keys["files"].items():
        obj = os.path.join(OBJ_DIR, s + ".o")
        objs.append(obj)
        keyfile = obj + ".key"
        have = open(keyfile).read().strip() if os.path.exists(keyfile) and os.path.exists(obj) else None
        if force or have != key:
            todo.append((s, obj, keyfile, key))

    postpass = postpass_enabled()
    postpass_stats = {}

    def compile_one(item):
        s, obj, keyfile, key = item
        src = os.path.join(CSRC, s)
        common = FLAGS + (extra_flags or []) + ["-I", os.path.join(ROOT, "include")]
        if not postpass:
            cmd = [hipcc] + common + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        else:
            # hipcc's own pipeline (hipcc -### -c), opened between the device compiler and the assembler: device assembly ->
            # asm_postpass -> assemble -> link the code object -> bundle -> host compile embedding the bundle
            from . import asm_postpass
            cuid = ["-cuid=" + hashlib.sha256(s.encode()).hexdigest()[:16]]      # the same compilation-unit id on both sides
            dev_s, dev_o, hsaco, hipfb = (obj + ext for ext in (".dev.s", ".dev.o", ".hsaco", ".hipfb"))
            steps = [
                [hipcc] + common + cuid + ["--cuda-device-only", "-S", src, "-o", dev_s],
                None,
                [os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}", "-c", dev_s, "-o", dev_o],
                [os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o],
                [os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
                 f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null", f"-input={hsaco}", f"-output={hipfb}"],
                [hipcc] + common + cuid + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", hipfb, "-c", src, "-o", obj],
            ]
            for cmd in steps:
                if cmd is None:
                    postpass_stats[s] = asm_postpass.rewrite_file(dev_s)
                    continue
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd, stderr=subprocess.DEVNULL if "--cuda-device-only" in cmd else None)
            for tmp in (dev_o, hsaco, hipfb):
                os.remove(tmp)
        with open(keyfile, "w") as f:
            f.write(key)

    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1), 8)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))
    idsrc = os.path.join(OBJ_DIR, "build_id.cpp")
    with open(idsrc, "w") as f:
        f.write('extern "C" const char* cs_build_id(void) { return "%s"; }\n' % keys["build_id"])
    idobj = os.path.join(OBJ_DIR, "build_id.o")
    subprocess.check_call(["g++", "-O1", "-fPIC", "-c", idsrc, "-o", idobj])
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + [idobj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if postpass_stats:
        keys["postpass"] = postpass_stats
        if verbose:
            tot = {k: sum(v[k] for v in postpass_stats.values()) for k in next(iter(postpass_stats.values()))}
            print("asm_postpass:", tot, flush=True)
    elif postpass and os.path.exists(MANIFEST_PATH):
        try:
            keys["postpass"] = json.load(open(MANIFEST_PATH)).get("postpass", {})
        except Exception:
            pass
    with open(MANIFEST_PATH, "w") as f:
        json.dump(keys, f, indent=1, sort_keys=True)
    last_action = f"compiled ({len(todo)} of {len(objs)} translation units)"
    return LIB_PATH


def _build_variant(variant: str, verbose: bool, extra_flags, jobs) -> str:
    global LIB_PATH, MANIFEST_PATH, OBJ_DIR, last_action
    saved = (LIB_PATH, MANIFEST_PATH, OBJ_DIR)
    try:
        LIB_PATH = os.path.join(PKG, f"libcrowdstep_{variant}.so")
        MANIFEST_PATH = LIB_PATH + ".manifest.json"
        OBJ_DIR = os.path.join(CSRC, f".build_{variant}")
        return build(force=False, verbose=verbose, extra_flags=(extra_flags or []) + [f"-DCS_VARIANT_{variant.upper()}=1"], jobs=jobs)
    finally:
        LIB_PATH, MANIFEST_PATH, OBJ_DIR = saved


last_action = "none"

if __name__ == "__main__":
    import sys

    var = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")]
    print(build(force="--force" in sys.argv, verbose=True, variant=var[0] if var else None), last_action)
