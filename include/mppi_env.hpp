// mppi_env.hpp -- stand-in plant for the closed loop.
//
// The reference steps a MuJoCo 2.0 scene (reference include/mppi_env.hpp:21-35,
// src/PointMassEnv.cpp:39-198); its vendored MuJoCo binaries need a personal licence key that
// expired in 2021 and are never loaded here.  This class keeps the interface the
// driver uses -- PointMassEnv(modelFile, mjkey, view), simulate(u) -> done, step(x, u), get_x(x),
// operator<< -- and integrates the same physics the MJCF files describe
// (envs/point_mass{1,2,3}d.xml): one slide joint per axis, no gravity, RK4 at `timestep`,
// joint damping and armature, motors with gear and control clamp, joint range.  The numbers are
// READ from the MJCF text when the file exists (plain text scan, no XML library, nothing
// executed); otherwise the documented defaults of those files are used.
// No viewer: `view` is accepted and ignored.
#ifndef MPPI_GPU_AMD_ENV_HPP_
#define MPPI_GPU_AMD_ENV_HPP_

#include <iostream>
#include <string>
#include <vector>

class Env {
public:
    virtual ~Env() {}
    friend std::ostream& operator<<(std::ostream& os, const Env& env) { return os << env.print(); }
    virtual std::string print() const { return info; }

protected:
    std::string info;
};

class PointMassEnv : public Env {
public:
    // mjkey is accepted for signature compatibility and never read
    PointMassEnv(const char* modelFile, const char* mjkey = nullptr, bool view = false);
    ~PointMassEnv() override {}
    std::string print() const override;

    // reference src/PointMassEnv.cpp:115-173: hold u, advance 1/60 s of plant time (two 0.01 s
    // steps), return true once the episode is over (plant time beyond ~10 s)
    bool simulate(float* u);
    // reference :175-187: one plant step under u, state out as positions then velocities
    void step(float* x, float* u);
    // reference :189-198
    void get_x(float* x);

    int dims() const { return (int)q_.size(); }
    double time() const { return time_; }
    void set_end_time(double t) { simend_ = t; }

private:
    void rk4(const std::vector<double>& ctrl);
    std::vector<double> q_, v_;
    double timestep_, damping_, armature_, gear_, ctrl_lo_, ctrl_hi_, range_lo_, range_hi_, mass_;
    double time_, simend_;
};

#endif  // MPPI_GPU_AMD_ENV_HPP_
