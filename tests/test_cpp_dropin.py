"""A C++ host written against the reference's `PointMassModel` interface builds with plain g++
against include/point_mass.hpp + libmppi_gpu_amd.so (CPU test), and on the GPU produces the same
numbers as the same run driven through the C ABI from Python (GPU test)."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "host_loop.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_loop")
LIBDIR = os.path.join(ROOT, "mppi_gpu_amd", "lib")


def _build():
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L", LIBDIR, "-lmppi_gpu_amd", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_reference_style_host_compiles_and_links_with_gpp():
    exe = _build()
    assert os.path.exists(exe)
    syms = subprocess.run(["nm", "-D", "-C", os.path.join(LIBDIR, "libmppi_gpu_amd.so")],
                          capture_output=True, text=True, check=True).stdout
    for member in ("PointMassModel::PointMassModel(int, int, float, int, int, bool)",
                   "PointMassModel::get_act(float*)",
                   "PointMassModel::memcpy_set_data(float*, float*, float*, float*)",
                   "PointMassModel::get_x(float*)", "PointMassModel::set_x(float*)",
                   "PointMassModel::get_u(float*)",
                   "PointMassModel::memcpy_get_data(float*, float*)",
                   "PointMassModel::get_inf(float*, float*, float*, float*, float*, float*, float*)",
                   "ControllerBase::ControllerBase(int, int, float, int, int)",
                   "ControllerBase::next(float const*, float*)"):
        assert member in syms, member


def test_headers_need_no_gpu_toolchain():
    """cost.hpp / point_mass_gpu.hpp / controller_base.hpp compile as plain host C++."""
    code = ('#include "cost.hpp"\n#include "point_mass_gpu.hpp"\n#include "controller_base.hpp"\n'
            '#include "point_mass.hpp"\n#include "mppi_gpu_amd.h"\n'
            'int main(){float w[2]={1,5},g[2]={1,0},inv[1]={1},x[2]={0.5f,0.25f},u[1]={0.1f},e[1]={0.01f};'
            'Cost c(w,2,g,2,1.0f,inv,1); float s=c.step_cost(x,u,e,0,0)+c.final_cost(x,0);'
            'float X[4]={0},x0[2]={0,0},U[1]={0.2f},E[1]={0.01f},xg[4]={1,0.1f,0,1},ug[2]={0.005f,0.1f};'
            'PointMassModelGpu m; m.init(X,x0,U,E,1,xg,2,ug,1,w,g,1.0f,0); s+=m.run(nullptr);'
            'return s>0?0:1;}')
    path = os.path.join(ROOT, "tests", "cpp", "_hdr_check.cpp")
    open(path, "w").write(code)
    exe = path[:-4]
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), path,
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert subprocess.run([exe]).returncode == 0
    os.remove(path); os.remove(exe)


@pytest.mark.gpu
def test_cpp_host_loop_equals_python_driven_run(gpu):
    from mppi_gpu_amd import PointMassModel
    exe = _build()
    K, T, iters = 3000, 50, 6
    env = dict(os.environ)
    out = subprocess.run([exe, str(K), str(T), str(iters)], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr + out.stdout
    acts = np.array([[float(a), float(b)] for a, b in
                     re.findall(r"ACT \d+ (\S+) (\S+)", out.stdout)], np.float32)
    assert acts.shape == (iters, 2)
    assert "Allocating Space... : Done" in out.stdout          # the reference's progress lines
    x = np.zeros(4, np.float32)
    dt = np.float32(0.1)
    with PointMassModel(K, T, 0.1, 4, 2) as m:
        m.set_seed(11)
        m.memcpy_set_data(x, np.zeros((T, 2), np.float32), [1, 0, 0, 0], [1, 1, 50, 50])
        for it in range(iters):
            a = m.get_act()
            assert np.array_equal(a, acts[it]), (it, a, acts[it])   # same library, same bits
            for i in range(2):
                p = x[i] + dt * x[i + 2] + np.float32(0.5) * dt * dt * a[i]
                v = x[i + 2] + dt * a[i]
                x[i], x[i + 2] = p, v
            m.set_x(x)
