// host_loop_sharded.cpp -- the reference's host loop (src/main.cu:309-374: ctor ->
// memcpy_set_data -> loop { get_u, get_act, plant step, set_x }) written against
// ShardedPointMassModel: ONE process, n shard engines (argv: samples horizon iters n_shards
// transport [same_device]).  With same_device = 1 every shard is placed on device 0 (rehearsal on a
// one-GPU box: transports direct and copy).  Compiled with plain g++; prints one line per step.
#include "point_mass_sharded.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2000;
    const int steps = argc > 2 ? atoi(argv[2]) : 50;
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const int shards = argc > 4 ? atoi(argv[4]) : 1;
    const char* transport = argc > 5 ? argv[5] : "collective";
    const bool same_device = argc > 6 && atoi(argv[6]) != 0;
    const int act_dim = 2, state_dim = 4;
    const float dt = 0.1f;

    std::vector<float> x(state_dim, 0.0f), u(steps * act_dim, 0.0f), next_act(act_dim);
    float goal[4] = {1, 0, 0, 0};            // reference config/point_mass2d.yaml
    float w[4] = {1, 1, 50, 50};
    std::vector<int> devs(shards, 0);

    ShardedPointMassModel* model = new ShardedPointMassModel(
        n, steps, dt, state_dim, act_dim, false, shards, transport,
        same_device ? devs.data() : nullptr);
    printf("SHARDS %d %s\n", model->n_shards(), model->transport());
    model->set_seed(11);
    model->memcpy_set_data(x.data(), u.data(), goal, w);

    std::vector<float> u_prev(steps * act_dim);
    for (int it = 0; it < iters; ++it) {
        model->get_u(u_prev.data());
        model->get_act(next_act.data());
        printf("ACT %d %.9g %.9g\n", it, next_act[0], next_act[1]);
        for (int a = 0; a < act_dim; ++a) {
            const float p = x[a] + dt * x[a + 2] + 0.5f * dt * dt * next_act[a];
            const float v = x[a + 2] + dt * next_act[a];
            x[a] = p;
            x[a + 2] = v;
        }
        model->set_x(x.data());
    }
    std::vector<float> xr(state_dim);
    model->get_x(xr.data());
    printf("X %.9g %.9g %.9g %.9g\n", xr[0], xr[1], xr[2], xr[3]);
    delete model;
    return 0;
}
