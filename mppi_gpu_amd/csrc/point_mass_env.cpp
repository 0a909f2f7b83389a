// point_mass_env.cpp -- stand-in plant (see include/mppi_env.hpp).  Host-only C++.
#include "../../include/mppi_env.hpp"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace {

// first numeric value of attribute `attr="..."` found after `anchor` in the MJCF text
bool scan_attr(const std::string& txt, const std::string& anchor, const std::string& attr, int index,
               double* out)
{
    // try every tag that contains the anchor until one carries the attribute
    for (size_t a = txt.find(anchor); a != std::string::npos; a = txt.find(anchor, a + 1)) {
        const size_t tag_end = txt.find('>', a);
        size_t k = txt.find(attr + "=\"", a);
        if (k == std::string::npos) return false;
        if (tag_end != std::string::npos && k > tag_end) continue;   // attribute of another tag
        k += attr.size() + 2;
        std::istringstream is(txt.substr(k, txt.find('"', k) - k));
        double v = 0;
        bool ok = true;
        for (int i = 0; i <= index; ++i)
            if (!(is >> v)) { ok = false; break; }
        if (!ok) continue;
        *out = v;
        return true;
    }
    return false;
}

int count_occurrences(const std::string& txt, const std::string& what)
{
    int n = 0;
    for (size_t p = txt.find(what); p != std::string::npos; p = txt.find(what, p + 1)) ++n;
    return n;
}

}  // namespace

PointMassEnv::PointMassEnv(const char* modelFile, const char* /*mjkey*/, bool /*view*/)
    : timestep_(0.01), damping_(0.1), armature_(0.01), gear_(10.0), ctrl_lo_(-1.0), ctrl_hi_(1.0),
      range_lo_(-1.4), range_hi_(1.4), mass_(0.0), time_(0.0), simend_(10.0)
{
    int n_axes = 2;
    double radius = 0.05;
    std::string txt;
    if (modelFile) {
        std::ifstream f(modelFile);
        if (f) {
            std::stringstream ss;
            ss << f.rdbuf();
            txt = ss.str();
        }
    }
    if (!txt.empty()) {
        scan_attr(txt, "<option", "timestep", 0, &timestep_);
        scan_attr(txt, "<joint armature", "armature", 0, &armature_);
        scan_attr(txt, "<joint armature", "damping", 0, &damping_);
        scan_attr(txt, "<motor ctrllimited", "ctrlrange", 0, &ctrl_lo_);
        scan_attr(txt, "<motor ctrllimited", "ctrlrange", 1, &ctrl_hi_);
        scan_attr(txt, "<motor gear", "gear", 0, &gear_);
        scan_attr(txt, "name=\"agent_x\"", "range", 0, &range_lo_);
        scan_attr(txt, "name=\"agent_x\"", "range", 1, &range_hi_);
        scan_attr(txt, "name=\"agent\" pos", "size", 0, &radius);
        const int n = count_occurrences(txt, "type=\"slide\"");
        if (n >= 1 && n <= 4) n_axes = n;
        info = std::string("PointMassEnv stand-in, model ") + modelFile;
    } else {
        // modelFile may also be just "1", "2" or "3": number of axes with the shipped defaults
        if (modelFile && modelFile[0] >= '1' && modelFile[0] <= '4' && modelFile[1] == 0)
            n_axes = modelFile[0] - '0';
        info = "PointMassEnv stand-in, default point-mass parameters";
    }
    // inertiafromgeom: sphere of default density 1000 kg/m^3
    mass_ = 1000.0 * 4.0 / 3.0 * 3.14159265358979323846 * radius * radius * radius;
    q_.assign(n_axes, 0.0);
    v_.assign(n_axes, 0.0);
}

std::string PointMassEnv::print() const
{
    std::ostringstream os;
    os << info << " (" << q_.size() << " slide axes, dt " << timestep_ << ", mass " << mass_
       << ", armature " << armature_ << ", damping " << damping_ << ", gear " << gear_ << ")";
    return os.str();
}

// (m + armature) * dv/dt = gear * clamp(u) - damping * v ; dq/dt = v ; classic RK4
void PointMassEnv::rk4(const std::vector<double>& ctrl)
{
    const double M = mass_ + armature_;
    const double h = timestep_;
    for (size_t i = 0; i < q_.size(); ++i) {
        const double f = gear_ * ctrl[i];
        auto acc = [&](double v) { return (f - damping_ * v) / M; };
        const double k1v = acc(v_[i]), k1q = v_[i];
        const double k2v = acc(v_[i] + 0.5 * h * k1v), k2q = v_[i] + 0.5 * h * k1v;
        const double k3v = acc(v_[i] + 0.5 * h * k2v), k3q = v_[i] + 0.5 * h * k2v;
        const double k4v = acc(v_[i] + h * k3v), k4q = v_[i] + h * k3v;
        q_[i] += h / 6.0 * (k1q + 2 * k2q + 2 * k3q + k4q);
        v_[i] += h / 6.0 * (k1v + 2 * k2v + 2 * k3v + k4v);
        // joint range as an inelastic stop (MuJoCo uses a soft constraint)
        if (q_[i] < range_lo_) { q_[i] = range_lo_; if (v_[i] < 0) v_[i] = 0; }
        if (q_[i] > range_hi_) { q_[i] = range_hi_; if (v_[i] > 0) v_[i] = 0; }
    }
    time_ += h;
}

bool PointMassEnv::simulate(float* u)
{
    if (time_ >= simend_) return true;
    std::vector<double> ctrl(q_.size());
    for (size_t i = 0; i < q_.size(); ++i)
        ctrl[i] = std::fmin(ctrl_hi_, std::fmax(ctrl_lo_, (double)u[i]));
    const double start = time_;
    while (time_ - start < 1.0 / 60.0) rk4(ctrl);
    return false;
}

void PointMassEnv::step(float* x, float* u)
{
    std::vector<double> ctrl(q_.size());
    for (size_t i = 0; i < q_.size(); ++i)
        ctrl[i] = std::fmin(ctrl_hi_, std::fmax(ctrl_lo_, (double)u[i]));
    rk4(ctrl);
    get_x(x);
}

void PointMassEnv::get_x(float* x)
{
    const size_t n = q_.size();
    for (size_t i = 0; i < n; ++i) {
        x[i] = (float)q_[i];
        x[i + n] = (float)v_[i];
    }
}
