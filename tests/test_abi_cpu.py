"""CPU suite: the C-ABI library loads without a GPU and exports every symbol include/crowdstep.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    from social_navigation_pyenvs_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "crowdstep.h")).read()
    declared = set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", header))
    declared -= {"cs_status"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libcrowdstep.so does not export {name}"
    assert set(_lib.ABI_SYMBOLS) == declared
    assert lib.cs_abi_version() == _lib.ABI_VERSION == 4
    hdr = open(os.path.join(ROOT, "include", "crowdstep.h")).read()
    assert "#define CS_ABI_VERSION 4" in hdr          # header, library and binding name one ABI


def test_library_on_disk_was_built_from_the_sources_on_disk():
    """The rebuild is keyed on a content hash of csrc/* + include/*: the id the LOADED library reports (cs_build_id) must
    be the hash of the sources in the tree, so a stale prebuilt .so cannot pass for the current code."""
    import ctypes as C

    import __graft_entry__ as g

    g.build()
    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.csrc import build as hb

    st = hb.status()
    assert st["exists"] and st["fresh"], st
    lib = _lib.load()
    lib.cs_build_id.restype = C.c_char_p
    assert lib.cs_build_id().decode() == st["build_id"]
    # and the key really follows the content: a changed source gives a different id
    keys = hb.source_keys()
    assert hb.source_keys(["-DX"])["build_id"] != keys["build_id"]


def test_struct_layout_matches_header():
    import ctypes as C

    from social_navigation_pyenvs_amd._lib import cs_worlds

    # 8 int32 + 7 pointers + 5 floats + 2 int32 (+4 padding) + 1 pointer + 1 int32 (+4 padding) + 1 pointer + 1 int32 (+4 padding)
    assert C.sizeof(cs_worlds) == 8 * 4 + 7 * 8 + 5 * 4 + 2 * 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4
    assert cs_worlds.d_state.offset == 32
    assert cs_worlds.orca_math.offset == C.sizeof(cs_worlds) - 8        # ABI 4: the ORCA arithmetic is a field of the context


def test_product_fails_loudly_without_gpu():
    from social_navigation_pyenvs_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import numpy as np

    from social_navigation_pyenvs_amd.batched import CrowdWorlds

    with pytest.raises(_lib.CrowdstepError):
        CrowdWorlds(np.zeros((1, 2, 13)), np.zeros((1, 2, 1, 2)), np.zeros((2, 20)), type=0)


def test_header_is_plain_c_and_generator_struct_matches():
    """include/crowdstep.h is the contract a non-C++ host binds: it must compile as C, and the ctypes mirrors of its
    structs must have the compiler's layout."""
    import ctypes as C
    import subprocess
    import tempfile

    from social_navigation_pyenvs_amd._lib import cs_gym_book, cs_stage_book, cs_worlds
    from social_navigation_pyenvs_amd.generators import cs_generator

    src = ('#include "crowdstep.h"\n#include <stdio.h>\n#include <stddef.h>\n'
           'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(cs_worlds), sizeof(cs_generator), '
           'offsetof(cs_generator, circle_radius), offsetof(cs_generator, robot_desired_speed), offsetof(cs_worlds, d_world_flags), offsetof(cs_worlds, d_orca_agent_params), '
           'sizeof(cs_gym_book), offsetof(cs_gym_book, clock_len), offsetof(cs_gym_book, d_reward), offsetof(cs_gym_book, seed_stride), '
           'sizeof(cs_stage_book), offsetof(cs_stage_book, d_failed), offsetof(cs_stage_book, depth));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes[0] == C.sizeof(cs_worlds)
    assert sizes[1] == C.sizeof(cs_generator)
    assert sizes[2] == cs_generator.circle_radius.offset
    assert sizes[3] == cs_generator.robot_desired_speed.offset
    assert sizes[4] == cs_worlds.d_world_flags.offset
    assert sizes[5] == cs_worlds.d_orca_agent_params.offset
    assert sizes[6] == C.sizeof(cs_gym_book) and sizes[7] == cs_gym_book.clock_len.offset and sizes[8] == cs_gym_book.d_reward.offset
    assert sizes[9] == cs_gym_book.seed_stride.offset
    assert sizes[10] == C.sizeof(cs_stage_book) and sizes[11] == cs_stage_book.d_failed.offset and sizes[12] == cs_stage_book.depth.offset


def test_integration_doc_stub_matches_the_struct():
    """The ctypes stub INTEGRATION.md shows a maintainer is the header's cs_worlds, field for field."""
    import ctypes as C

    from social_navigation_pyenvs_amd._lib import cs_worlds

    src = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"class cs_worlds\(C\.Structure\):.*?\n(    _fields_ = \[.*?\])\n", src, re.S)
    assert m, "stub not found"
    ns = {"C": C}
    exec("class stub(C.Structure):\n" + m.group(1), ns)
    stub = ns["stub"]
    assert [(f[0], f[1]) for f in stub._fields_] == [(f[0], f[1]) for f in cs_worlds._fields_]
    assert C.sizeof(stub) == C.sizeof(cs_worlds)


def test_hip_runtime_preload_checks_the_soname_and_maps_one_runtime(tmp_path, monkeypatch):
    """_lib.load() preloads torch's bundled libamdhip64 only when its SONAME is the one libcrowdstep.so NEEDs (so that the loader
    binds the library to it and a later `import torch` finds the same file mapped): exactly ONE libamdhip64 in /proc/self/maps here;
    a bundled runtime with another SONAME is not preloaded (a warning, the system runtime instead of two mapped runtimes)."""
    import subprocess
    import sys

    from social_navigation_pyenvs_amd import _lib

    _lib.load()
    soname, needed = _lib._elf_dynamic(_lib.LIB_PATH)
    want = [x for x in needed if x.startswith("libamdhip64")]
    assert len(want) == 1
    if _lib.hip_runtime_path:
        assert _lib._elf_dynamic(_lib.hip_runtime_path)[0] == want[0]
    assert len(_lib.mapped_hip_runtimes()) == 1, _lib.mapped_hip_runtimes()
    # a bundled runtime whose SONAME differs: simulated by pointing the check at a library that is not a HIP runtime at all
    code = ("import warnings, sys; sys.path.insert(0, %r)\n"
            "from social_navigation_pyenvs_amd import _lib\n"
            "real = _lib._elf_dynamic\n"
            "_lib._elf_dynamic = lambda p: ('libamdhip64.so.6', []) if 'torch' in p else real(p)\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always'); _lib.load()\n"
            "assert _lib.hip_runtime_path is None and any('not preloading' in str(x.message) for x in w), (_lib.hip_runtime_path, [str(x.message) for x in w])\n"
            "assert len(_lib.mapped_hip_runtimes()) == 1\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    subprocess.check_call([sys.executable, "-c", code])


def test_new_entry_points_reject_bad_arguments_before_touching_a_device():
    """Argument checks of the round-4 entry points come first: null pointers, a staging depth that is not a power of two, a staging
    batch of the wrong size, a generator that does not fit the worlds -- all CS_ERR_ARG (ValueError through the binding) with no GPU."""
    import ctypes as C

    from social_navigation_pyenvs_amd import _lib
    from social_navigation_pyenvs_amd.generators import cs_generator

    lib = _lib.load()
    null = C.c_void_p(None)
    assert lib.cs_gym_step(null, C.c_float(0.0125), C.c_int(20), null, C.c_float(0.25), null, null, null, null, C.c_int(0), null, null) == _lib.CS_ERR_ARG
    assert lib.cs_refill_staged_worlds(null, null, null, null) == _lib.CS_ERR_ARG
    assert lib.cs_consume_staged_worlds(null, null, null, null, null, C.c_int(0), null, null) == _lib.CS_ERR_ARG
    buf = C.create_string_buffer(8)
    assert lib.cs_device_pci_bus_id(C.c_int(0), buf, C.c_size_t(8)) == _lib.CS_ERR_ARG          # buffer too small for "0000:00:00.0"
    done = C.c_int(0)
    assert lib.cs_event_query(null, C.byref(done)) == _lib.CS_ERR_ARG
    # a well-formed descriptor pair with a bad stage book
    w = _lib.cs_worlds(W=8, n=5, G=2, type=3, layout=_lib.CS_LAYOUT_AOS, d_state=1, d_goals=1)
    st = _lib.cs_worlds(W=24, n=5, G=2, type=3, layout=_lib.CS_LAYOUT_AOS, d_state=1, d_goals=1)
    g = cs_generator(scenario=0, n=5, insert_robot=1, randomize_attributes=0, randomize_positions=1, max_tries=100, circle_radius=7.0,
                     traffic_length=14.0, traffic_height=3.0, robot_radius=0.3, human_mass=75.0, robot_mass=80.0, robot_desired_speed=1.0)
    book = _lib.cs_stage_book(d_seeds=1, d_base_seed=1, d_epoch=1, d_staged_seed=1, d_staged_status=1, d_failed=1, seed_stride=0, depth=3)
    assert lib.cs_refill_staged_worlds(C.byref(g), C.byref(st), C.byref(book), null) == _lib.CS_ERR_ARG   # depth 3: not a power of two
    assert "power of two" in lib.cs_last_error().decode()
    book.depth = 4
    mask = C.c_void_p(1)
    assert lib.cs_consume_staged_worlds(C.byref(g), C.byref(st), C.byref(w), mask, C.byref(book), C.c_int(0), null, null) == _lib.CS_ERR_ARG
    assert "differ in shape" in lib.cs_last_error().decode()                                             # 24 / 4 = 6 worlds staged per level, 8 live
    g.n = 7
    st.W = 32
    assert lib.cs_refill_staged_worlds(C.byref(g), C.byref(st), C.byref(book), null) == _lib.CS_ERR_ARG
    assert "cs_generator.n differs" in lib.cs_last_error().decode()
