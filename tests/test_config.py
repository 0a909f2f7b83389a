"""Configuration reader: the reference's YAML keys, checked with the values its own
parse self-check asserts (reference src/main.cu:686-725, verify_parse, against its
config/mppi-config-test.yaml): n=3, state 4, act 2, horizon 12, dt .1, lambda 1.5,
max-a [1.2,1.3], noise [.24,.26], init-act [.1,.2], cost.w [1,2,.5,.75], goal [1,2,3,4]."""
import os
import subprocess

from conftest import ROOT

TEST_YAML = """---
    action-dim: 2
    cost:
      type: quadratic
      w:
        - 1
        - 2
        - 0.5
        - 0.75
    dt: 0.1
    env: ../envs/point_mass.xml
    goal:
      - 1
      - 2
      - 3
      - 4
    horizon: 12
    init-act:
      - 0.1
      - 0.2
    lambda: 1.5
    max-a:
      - 1.2
      - 1.3
    noise:
      - 0.24
      - 0.26
    samples: 3
    state-dim: 4
"""


def _exe(tmp_path):
    exe = str(tmp_path / "config_check")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "config_check.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_verify_parse_known_answer(tmp_path):
    exe = _exe(tmp_path)
    p = tmp_path / "cfg.yaml"
    p.write_text(TEST_YAML)
    out = subprocess.run([exe, str(p)], capture_output=True, text=True, check=True).stdout
    lines = dict(l.split(" ", 1) if " " in l else (l, "") for l in out.strip().splitlines()[1:])
    head = out.splitlines()[0]
    assert "samples=3 state=4 act=2 horizon=12 dt=0.100000001 lambda=1.5 type=quadratic" in head
    assert "env=../envs/point_mass.xml" in head
    tol = 1e-6                                            # the reference's TOL
    def close(name, exp):
        got = [float(v) for v in lines[name].split()]
        assert len(got) == len(exp) and all(abs(a - b) < tol for a, b in zip(got, exp)), (name, got)
    close("max_a", [1.2, 1.3]); close("noise", [0.24, 0.26]); close("init", [0.1, 0.2])
    close("w", [1, 2, 0.5, 0.75]); close("goal", [1, 2, 3, 4])
    assert lines["consistent=1"] == "" or "consistent=1" in out


def test_missing_key_is_an_error(tmp_path):
    exe = _exe(tmp_path)
    p = tmp_path / "bad.yaml"
    p.write_text(TEST_YAML.replace("    lambda: 1.5\n", ""))
    r = subprocess.run([exe, str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "missing key: lambda" in r.stdout
