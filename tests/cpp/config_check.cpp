#include "mppi_config.hpp"
#include <cstdio>
int main(int argc, char** argv)
{
    MppiConfig c;
    if (argc < 2 || !c.parse_file(argv[1])) { printf("ERROR %s\n", c.error.c_str()); return 1; }
    printf("env=%s samples=%d state=%d act=%d horizon=%d dt=%.9g lambda=%.9g type=%s\n", c.env.c_str(),
           c.samples, c.state_dim, c.act_dim, c.horizon, c.dt, c.lambda, c.cost_type.c_str());
    auto pl = [](const char* n, const std::vector<float>& v) { printf("%s", n); for (float f : v) printf(" %.9g", f); printf("\n"); };
    pl("noise", c.noise); pl("init", c.init_act); pl("max_a", c.max_a); pl("goal", c.goal); pl("w", c.cost_w);
    printf("consistent=%d\n", (int)c.consistent());
    return 0;
}
