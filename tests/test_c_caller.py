"""The C-ABI from plain C (tests/c/abi_smoke.c): the header must be valid C99 on its own (what a foreign-language
binding generator sees), and a C program linked against libqpn_hip.so gets the hand-derived known answers of
SURVEY.md section 8(c) through qpn_solve_mcp_csc (PATHSolver.solve_mcp's argument list) and qpn_solve_nodes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "quadraticprogramnetworks.jl_amd")


def test_header_is_plain_c99(tmp_path):
    obj = str(tmp_path / "abi_smoke.o")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{INC}", "-c", SRC, "-o", obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_c_program_gets_the_known_answers(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", f"-I{INC}", SRC, "-o", exe, f"-L{LIBDIR}", "-lqpn_hip", "-lm", f"-Wl,-rpath,{LIBDIR}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout + r.stderr
