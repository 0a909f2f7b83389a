// mppi_closed_loop.cpp -- the closed-loop driver of the reference (src/main.cu:220-399) on the
// MI355X engine and the stand-in plant: get_u -> get_act (timed) -> env.simulate -> env.get_x ->
// set_x, until the plant's episode ends; prints the reference's "Average controller execution
// time" and writes the trajectory CSV in the reference's column format (src/main.cu:32-57) for
// 2-D, generalised to A axes.  BASELINE config 5: point_mass3d, K=1e5, T=200, 100 Hz re-plan ->
// the solve must fit 10 ms.
//
//   mppi_closed_loop [-c config.yaml] [-k key] [--dims A] [--samples K] [--horizon T] [--dt 0.1]
//                    [--model file.xml] [--seconds S] [--rate-hz R] [-t|--traj-save out.csv] [-s|--step-save prefix]
//                    [--lambda L] [--noise SIGMA] [--max-a config|LIMIT]
//                    [--gpus N [--transport collective|direct|copy]]
// --gpus N (N >= 1, or `all`) runs the controller over N GPUs of this process through
// ShardedPointMassModel (include/point_mass_sharded.hpp): one shard engine and host thread per
// device, the per-solve exchange by RCCL all-gather (default), peer stores or peer copies.  Without
// it the single-GPU PointMassModel is used, as in the reference (src/main.cu:311).
//                    [--use-config-params]
// -c reads a configuration file with the reference's keys (include/mppi_config.hpp); options
// given after it override single values (the reference's -c/--config, src/main.cu:401-453).
// Like the reference, the file's `lambda` and `init-act` are parsed and NOT applied (its controller
// runs with lambda 1 and zero initial controls whatever the file says);
// --use-config-params applies them.
// -k/--key (the reference's MuJoCo licence file, src/main.cu:417-423) is accepted and ignored: the
// stand-in plant needs no key.  --max-a switches the action limit ON (the reference parses max-a,
// src/main.cu:524,566-568, and never applies it, so the default is off): `config` takes the
// file's max-a list, a number limits every axis to +-LIMIT.
#include "mppi_config.hpp"
#include "mppi_env.hpp"
#include "point_mass.hpp"
#include "point_mass_sharded.hpp"

#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

static void to_csv_traj(const std::string& filename, const std::vector<std::vector<float>>& x,
                        const std::vector<std::vector<float>>& u, int A)
{
    std::ofstream out(filename);
    const char* ax[4] = {"x", "y", "z", "w"};
    for (int i = 0; i < A; ++i) out << ax[i] << ",";
    for (int i = 0; i < A; ++i) out << "v" << ax[i] << ",";
    for (int i = 0; i < A; ++i) out << "u" << ax[i] << ",";
    out << "size_x,size_u\n";
    for (size_t r = 0; r < x.size(); ++r) {
        for (float v : x[r]) out << v << ",";
        if (r < u.size())
            for (float v : u[r]) out << v << ",";
        if (r == 0) out << x.size() << "," << u.size();
        out << "\n";
    }
}

// per-step dump of the reference's save_step mode (src/main.cu:90-156, to_csv2; -s/--step-save),
// same column order, generalised from the reference's hard-coded 2-D columns to A axes:
// sample, states (positions, velocities), e, u[d], u_prev[d] (first sample's rows only), c, w
static void to_csv_step(const std::string& filename, const float* x, const float* u,
                        const float* u_prev, const float* e, const float* cost, const float* wts,
                        int sample, int size, int s_dim, int a_dim)
{
    std::ofstream out(filename);
    const char* ax[4] = {"x", "y", "z", "w"};
    out << "sample";
    for (int d = 0; d < a_dim; ++d) out << "," << ax[d];
    for (int d = 0; d < a_dim; ++d) out << "," << ax[d] << "_dot";
    for (int d = 0; d < a_dim; ++d) out << ",e_" << ax[d];
    for (int d = 0; d < a_dim; ++d) out << ",u[" << d << "]";
    for (int d = 0; d < a_dim; ++d) out << ",u_prev[" << d << "]";
    out << ",c,w\n";
    for (int i = 0; i < sample; ++i) {
        for (int j = 0; j < size + 1; ++j) {
            out << i;
            for (int d = 0; d < s_dim; ++d) out << "," << x[((size_t)i * (size + 1) + j) * s_dim + d];
            for (int d = 0; d < a_dim; ++d) {
                out << ",";
                if (j < size) out << e[((size_t)i * size + j) * a_dim + d];
            }
            for (int d = 0; d < a_dim; ++d) {
                out << ",";
                if (i < 1 && j < size) out << u[j * a_dim + d];
            }
            for (int d = 0; d < a_dim; ++d) {
                out << ",";
                if (i < 1 && j < size) out << u_prev[j * a_dim + d];
            }
            // the reference lists cost and weight of sample number (row index) on the first rows
            const size_t row = (size_t)i * (size + 1) + j;
            if (row < (size_t)sample) out << "," << cost[row] << "," << wts[row];
            out << "\n";
        }
    }
}

int main(int argc, char** argv)
{
    int A = 3, K = 100000, T = 200;
    float dt = 0.1f, lambda = 1.0f, sigma = 0.025f;
    double seconds = 2.0, rate_hz = 0.0;
    std::string model, traj, step_prefix;
    std::vector<float> cfg_goal, cfg_w, cfg_init, cfg_max_a;
    std::string max_a_opt, transport = "collective";
    int gpus = -1;              // -1: single-GPU PointMassModel; 0: all visible; n: n shards
    bool use_cfg_params = false, lambda_given = false;
    float cfg_lambda = 0.f;
    for (int i = 1; i < argc; i += 2) {
        std::string k = argv[i];
        if (k == "--use-config-params") { use_cfg_params = true; --i; continue; }   // a flag: no value
        if (i + 1 >= argc) { fprintf(stderr, "option %s needs a value\n", k.c_str()); return 2; }
        std::string v = argv[i + 1];
        if (k == "-c" || k == "--config") {
            MppiConfig cfg;
            if (!cfg.parse_file(v) || !cfg.consistent()) {
                fprintf(stderr, "config error: %s\n", cfg.error.c_str());
                return 1;
            }
            A = cfg.act_dim; K = cfg.samples; T = cfg.horizon; dt = cfg.dt; cfg_lambda = cfg.lambda;
            // the shipped files carry noise 0.25 while the reference's effective sigma is its
            // hard-coded 0.025; --noise overrides explicitly
            cfg_goal = cfg.goal; cfg_w = cfg.cost_w; cfg_init = cfg.init_act; cfg_max_a = cfg.max_a;
            if (model.empty()) model = cfg.env;
        } else if (k == "-k" || k == "--key") { /* licence key of the reference's MuJoCo: unused */ }
        else if (k == "--max-a") max_a_opt = v;
        else if (k == "--dims") A = atoi(v.c_str());
        else if (k == "--samples") K = atoi(v.c_str());
        else if (k == "--horizon") T = atoi(v.c_str());
        else if (k == "--dt") dt = (float)atof(v.c_str());
        else if (k == "--model") model = v;
        else if (k == "--seconds") seconds = atof(v.c_str());
        else if (k == "--rate-hz") rate_hz = atof(v.c_str());      // control steps per second of wall time (0: as fast as it goes)
        else if (k == "--traj" || k == "--traj-save" || k == "-t") traj = v;
        else if (k == "--step-save" || k == "-s") step_prefix = v;
        else if (k == "--lambda") { lambda = (float)atof(v.c_str()); lambda_given = true; }
        else if (k == "--noise") sigma = (float)atof(v.c_str());
        else if (k == "--gpus") gpus = (v == "all") ? 0 : atoi(v.c_str());
        else if (k == "--transport") transport = v;
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    // The reference parses `lambda` and `init-act` and never hands them to its controller (effective
    // values 1 and 0: src/main.cu:311, src/point_mass.cu:53-54, src/main.cu:678-684):
    // they are applied only on request.  --lambda on the command line always counts.
    if (use_cfg_params && cfg_lambda > 0.f && !lambda_given) lambda = cfg_lambda;
    if (!use_cfg_params) cfg_init.clear();
    if (A < 1 || A > 4) { fprintf(stderr, "--dims must be 1..4, got %d\n", A); return 2; }
    if (K < 1 || T < 1) { fprintf(stderr, "--samples and --horizon must be >= 1\n"); return 2; }
    const int S = 2 * A;
    // goal / cost.w of the shipped configs (reference config/point_mass{1,2,3}d.yaml)
    const float goals[4][8] = {{1, 0}, {1, 0, 0, 0}, {1, .5f, .75f, 0, 0, 0}, {1, .5f, .75f, .25f}};
    const float ws[4][8] = {{1, 5}, {1, 1, 50, 50}, {1, 1, 1, 5, 5, 5}, {1, 1, 1, 1, 5, 5, 5, 5}};
    std::vector<float> goal(goals[A - 1], goals[A - 1] + S), w(ws[A - 1], ws[A - 1] + S);
    if (!cfg_goal.empty()) { goal = cfg_goal; w = cfg_w; }

    std::string axes(1, (char)('0' + A));
    PointMassEnv env(model.empty() ? axes.c_str() : model.c_str(), nullptr, false);
    env.set_end_time(seconds);
    std::cout << env << std::endl;
    if (env.dims() != A) { fprintf(stderr, "model has %d axes, --dims %d\n", env.dims(), A); return 2; }

    printf("controller parameters: lambda %g sigma %g init-act %s\n", lambda, sigma,
           cfg_init.empty() ? "zero" : "from config");
    fflush(stdout);
    std::vector<float> x(S);
    double ctl_ms = 0.0, worst_ms = 0.0;
    size_t t = 0;
    std::vector<std::vector<float>> xs, us;
    auto drive = [&](auto* model_ctl) -> int {
        std::vector<float> sig(A, sigma);
        model_ctl->set_params(lambda, sig.data(), nullptr);
        if (!max_a_opt.empty()) {
            std::vector<float> lim(A, (float)atof(max_a_opt.c_str()));
            if (max_a_opt == "config") {
                if ((int)cfg_max_a.size() != A) { fprintf(stderr, "--max-a config needs -c with max-a\n"); return 2; }
                lim = cfg_max_a;
            }
            model_ctl->set_action_limit(lim.data());
            printf("action limit on:");
            for (float v : lim) printf(" %g", v);
            printf("\n");
        }
        std::vector<float> U(T * A, 0.0f), next_act(A), u_prev(T * A);
        if (!cfg_init.empty())          // reference init_action_seq, src/main.cu:678-684
            for (int t = 0; t < T; ++t)
                for (int a = 0; a < A; ++a) U[t * A + a] = cfg_init[a];
        env.get_x(x.data());
        model_ctl->memcpy_set_data(x.data(), U.data(), goal.data(), w.data());

        xs.push_back(x);
        bool done = false;
        const auto t_loop0 = std::chrono::steady_clock::now();
        while (!done) {
            model_ctl->get_u(u_prev.data());
            auto t1 = std::chrono::steady_clock::now();
            model_ctl->get_act(next_act.data());
            auto t2 = std::chrono::steady_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
            ctl_ms += ms;
            if (t > 0 && ms > worst_ms) worst_ms = ms;      // first call includes one-off set-up
            done = env.simulate(next_act.data());
            env.get_x(x.data());
            us.push_back(next_act);
            xs.push_back(x);
            if (!step_prefix.empty()) {          // reference src/main.cu:355-366 (save_step)
                std::vector<float> hx((size_t)K * (T + 1) * S), hu((size_t)T * A), he((size_t)K * T * A),
                    hc(K), hw(K);
                float beta = 0, nabla = 0;
                model_ctl->get_inf(hx.data(), hu.data(), he.data(), hc.data(), &beta, &nabla, hw.data());
                to_csv_step(step_prefix + std::to_string(t), hx.data(), hu.data(), u_prev.data(),
                            he.data(), hc.data(), hw.data(), K, T, S, A);
            }
            model_ctl->set_x(x.data());
            ++t;
            if (rate_hz > 0.0)          // a plant that runs in real time: the next step is due at t / rate
                std::this_thread::sleep_until(t_loop0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                            std::chrono::duration<double>((double)t / rate_hz)));
        }
        return 0;
    };
    int rc_drive;
    if (gpus >= 0) {
        if (transport != "collective" && transport != "direct" && transport != "copy") {
            fprintf(stderr, "--transport must be collective, direct or copy\n");
            return 2;
        }
        ShardedPointMassModel ctl(K, T, dt, S, A, false, gpus, transport.c_str());
        printf("controller: %d shard(s), transport %s\n", ctl.n_shards(), ctl.transport());
        rc_drive = drive(&ctl);
    } else {
        PointMassModel ctl(K, T, dt, S, A, false);
        rc_drive = drive(&ctl);
    }
    if (rc_drive) return rc_drive;
    double dist = 0;
    for (int i = 0; i < A; ++i) dist += (x[i] - goal[i]) * (x[i] - goal[i]);
    std::cout << "Average controller execution time: " << ctl_ms / t << std::endl;
    std::cout << "T: " << t << std::endl;
    std::cout << "Delta: " << ctl_ms << std::endl;
    printf("RESULT steps=%zu avg_ms=%.4f worst_ms=%.4f final_dist=%.4f budget_ms=10\n", t,
           ctl_ms / t, worst_ms, std::sqrt(dist));
    if (!traj.empty()) to_csv_traj(traj, xs, us, A);
    return 0;
}
