// prints the stand-in plant's response for tests/test_closed_loop.py
#include "mppi_env.hpp"
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv)
{
    PointMassEnv env(argc > 1 ? argv[1] : "2", nullptr, false);
    std::cout << env << std::endl;
    const int n = env.dims();
    float u[4] = {0.5f, -2.0f, 0.25f, 0.f}, x[8];
    for (int i = 0; i < 3; ++i) { env.step(x, u); }
    printf("STEP3");
    for (int i = 0; i < 2 * n; ++i) printf(" %.9g", x[i]);
    printf("\n");
    int frames = 0;
    env.set_end_time(1.0);
    while (!env.simulate(u)) ++frames;
    env.get_x(x);
    printf("FRAMES %d TIME %.6f\nEND", frames, env.time());
    for (int i = 0; i < 2 * n; ++i) printf(" %.9g", x[i]);
    printf("\n");
    return 0;
}
