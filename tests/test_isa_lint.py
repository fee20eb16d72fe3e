"""Build-time guard of the hand-placed prefetch idiom (tools/isa_lint.py): in the compiled gfx950 ISA of every kernel no
instruction may touch the destination VGPR of an inline-asm `global_load_*` before the hand-placed `s_waitcnt` that
covers it.  hipcc cross-compiles the listings here; no GPU needed."""
import importlib.util
import os
import subprocess
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "marl-ctf-development_amd", "csrc")

spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)


def _listing(tmp_path, body):
    path = tmp_path / "k.s"
    path.write_text(textwrap.dedent("""\
        \t.type\tk_demo,@function
        k_demo:
        """) + textwrap.dedent(body) + ".Lfunc_end0:\n")
    return str(path)


def test_lint_flags_a_copy_of_an_in_flight_register_at_a_back_edge(tmp_path):
    # the round-1 failure shape: the load's destination is copied at the loop's back edge, the wait is at the top of the next pass
    path = _listing(tmp_path, """\
        \ts_mov_b32 s0, 0
        .LBB0_1:
        \t;;#ASMSTART
        \ts_waitcnt vmcnt(8)
        \t;;#ASMEND
        \tv_add_u32_e32 v3, v5, v5
        \t;;#ASMSTART
        \tglobal_load_dword v7, v[0:1], off
        \t;;#ASMEND
        \tglobal_store_dword v[0:1], v3, off
        \tv_mov_b32_e32 v5, v7
        \ts_cbranch_scc0 .LBB0_1
        \ts_endpgm
        """)
    summary, bad = isa_lint.lint_file(path)
    assert summary == {"k_demo": 1}
    assert [(b[2], b[3]) for b in bad] == [("v_mov_b32_e32 v5, v7", [7])]


def test_lint_accepts_a_wait_in_the_issuing_iteration_and_sees_register_ranges(tmp_path):
    ok = _listing(tmp_path, """\
        .LBB0_1:
        \t;;#ASMSTART
        \tglobal_load_dwordx2 v[6:7], v[0:1], off
        \t;;#ASMEND
        \tglobal_store_dword v[0:1], v3, off
        \t;;#ASMSTART
        \ts_waitcnt vmcnt(1)
        \t;;#ASMEND
        \tv_mov_b32_e32 v5, v7
        \ts_cbranch_scc0 .LBB0_1
        \ts_endpgm
        """)
    assert isa_lint.lint_file(ok)[1] == []
    bad = _listing(tmp_path, """\
        \t;;#ASMSTART
        \tglobal_load_dwordx2 v[6:7], v[0:1], off
        \t;;#ASMEND
        \tglobal_store_dwordx4 v[0:1], v[4:7], off
        \ts_waitcnt vmcnt(0)
        \ts_endpgm
        """)
    assert [b[3] for b in isa_lint.lint_file(bad)[1]] == [[6, 7]]


@pytest.mark.parametrize("target,listing,users", [("asm", "ctf_kernels.s", ("k_observeILi16E", "k_observeILi4E", "k_observeILi1E")),
                                                  ("asm-policy", "ctf_policy.s", ("k_policy_featuresILi11E", "k_policy_featuresILi15E",
                                                                                  "k_policy_features_teamILi11E", "k_policy_features_teamILi15E")),
                                                  ("asm-policy-fact", "ctf_policy_fact.s", ("k_policy_features_factILi11E", "k_policy_features_factILi15E"))])
def test_shipped_kernels_keep_their_prefetch_registers_untouched(target, listing, users):
    """The idiom's users: the wave-per-env render (k_observe: next env's record / grid dword) and both policy front kernels
    (G = 11 and 15).  k_observe_tiles, k_step and the rest load through the compiler."""
    subprocess.check_call(["make", "-C", CSRC, "-s", target], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    summary, bad = isa_lint.lint_file(os.path.join(CSRC, listing))
    assert summary, "no hand-placed loads found: the listing or the parser changed"
    for u in users:
        assert any(u in k for k in summary), f"{u}: the hand-placed loads were not recognised"
    assert bad == [], "\n".join(f"{k}:{l}: `{t}` touches in-flight v{r}" for k, l, t, r in bad)
    n_kernels = len(isa_lint.parse_functions(os.path.join(CSRC, listing)))
    assert n_kernels >= (6 if target != "asm-policy-fact" else 5), "the listing was not parsed into its kernels"
