// point_mass.cpp -- the C++ veneer `PointMassModel` over the C ABI.
// Mirrors the reference's host class (src/point_mass.cu:19-491) call for call; the progress
// lines the reference prints from its constructor and memcpy_set_data
// (src/point_mass.cu:57,105,210,226) are kept, the per-stage device synchronisations are not
// (a solve is two stream-ordered launches, see engine.hip).
#include "../../include/point_mass.hpp"

#include "../../include/mppi_gpu_amd.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

// reference include/mppi_utils.hpp:19-25 (CUDA_CALL_CONST): print file:line:code, exit(1)
#define MPPI_CALL_CONST(x)                                                           \
    do {                                                                             \
        int err__ = (x);                                                             \
        if (err__ != MPPI_OK) {                                                      \
            printf("API error failed %s:%d Returned: %d (%s)\n", __FILE__, __LINE__, \
                   err__, mppi_last_error());                                        \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

PointMassModel::PointMassModel(int nb_sim, int steps, float dt, int state_dim, int act_dim,
                               bool verbose)
    : engine_(nullptr)
{
    std::cout << "Allocating Space... : " << std::flush;
    MPPI_CALL_CONST(mppi_create(nb_sim, steps, dt, state_dim, act_dim, verbose ? 1 : 0, &engine_));
    std::cout << "Done" << std::endl;
}

PointMassModel::~PointMassModel() { mppi_destroy(engine_); }

void PointMassModel::get_act(float* next_act) { MPPI_CALL_CONST(mppi_get_act(engine_, next_act)); }

void PointMassModel::memcpy_set_data(float* x, float* u, float* goal, float* w)
{
    std::cout << "Setting inital state of the sims... : " << std::flush;
    MPPI_CALL_CONST(mppi_set_data(engine_, x, u, goal, w));
    std::cout << "Done" << std::endl;
}

void PointMassModel::get_x(float* x) { MPPI_CALL_CONST(mppi_get_x(engine_, x)); }

void PointMassModel::memcpy_get_data(float* x_all, float* e)
{
    MPPI_CALL_CONST(mppi_get_data(engine_, x_all, e));
}

void PointMassModel::get_inf(float* x, float* u, float* e, float* cost, float* beta, float* nabla,
                             float* weight)
{
    std::cout << "Collect informations: " << std::endl;
    MPPI_CALL_CONST(mppi_get_inf(engine_, x, u, e, cost, beta, nabla, weight));
}

void PointMassModel::set_x(float* x) { MPPI_CALL_CONST(mppi_set_x(engine_, x)); }
void PointMassModel::get_u(float* u) { MPPI_CALL_CONST(mppi_get_u(engine_, u)); }

void PointMassModel::set_params(float lambda, const float* sigma, const float* inv_s)
{
    MPPI_CALL_CONST(mppi_set_params(engine_, lambda, sigma, inv_s));
}
void PointMassModel::set_seed(unsigned long long seed) { MPPI_CALL_CONST(mppi_set_seed(engine_, seed)); }
void PointMassModel::set_noise(const float* e) { MPPI_CALL_CONST(mppi_set_noise(engine_, e)); }
void PointMassModel::set_ref_compat(bool on) { MPPI_CALL_CONST(mppi_set_ref_compat(engine_, on)); }
void PointMassModel::set_noise_store(bool on) { MPPI_CALL_CONST(mppi_set_noise_store(engine_, on)); }
void PointMassModel::set_action_limit(const float* max_a)
{
    MPPI_CALL_CONST(mppi_set_action_limit(engine_, max_a));
}
void PointMassModel::set_tuning(int chunks, bool strict, int max_blocks)
{
    MPPI_CALL_CONST(mppi_set_tuning(engine_, chunks, strict ? 1 : 0, max_blocks));
}
