// mppi_config.hpp -- reader for the reference's controller configuration files
// (reference config/point_mass{1,2,3}d.yaml, parsed there with yaml-cpp in
// src/main.cu:455-628).  Same keys: env, samples, state-dim, action-dim, horizon, dt, lambda,
// noise[], init-act[], max-a[], goal[], cost.type, cost.w[].  yaml-cpp is not available in this
// image, so this is a header-only reader for exactly the YAML subset those files use
// (`key: scalar`, block sequences of scalars, one level of nesting for `cost:`); a missing key
// is an error, like in the reference (message + false instead of exit(1)).
//
// Unlike the reference the parsed lambda and noise can actually be forwarded to
// the controller: MppiConfig::apply(PointMassModel&).
#ifndef MPPI_GPU_AMD_CONFIG_HPP_
#define MPPI_GPU_AMD_CONFIG_HPP_

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct MppiConfig {
    std::string env, cost_type;
    int samples = 0, state_dim = 0, act_dim = 0, horizon = 0;
    float dt = 0.f, lambda = 0.f;
    std::vector<float> noise, init_act, max_a, goal, cost_w;
    std::string error;

    static std::string trim(const std::string& s)
    {
        const size_t a = s.find_first_not_of(" \t\r\n");
        if (a == std::string::npos) return "";
        const size_t b = s.find_last_not_of(" \t\r\n");
        return s.substr(a, b - a + 1);
    }

    bool parse_text(const std::string& text)
    {
        std::map<std::string, std::string> scalars;
        std::map<std::string, std::vector<float>> lists;
        std::istringstream in(text);
        std::string line, cur_list, parent;
        int parent_indent = -1;
        while (std::getline(in, line)) {
            const size_t hash = line.find('#');
            if (hash != std::string::npos) line = line.substr(0, hash);
            const std::string t = trim(line);
            if (t.empty() || t == "---") continue;
            const int indent = (int)line.find_first_not_of(" \t");
            if (t[0] == '-') {                              // sequence item of the open list
                if (cur_list.empty()) { error = "list item without a key: " + t; return false; }
                lists[cur_list].push_back((float)atof(trim(t.substr(1)).c_str()));
                continue;
            }
            const size_t colon = t.find(':');
            if (colon == std::string::npos) { error = "cannot parse line: " + t; return false; }
            std::string key = trim(t.substr(0, colon));
            const std::string val = trim(t.substr(colon + 1));
            if (parent_indent >= 0 && indent <= parent_indent) { parent.clear(); parent_indent = -1; }
            if (!parent.empty()) key = parent + "." + key;
            if (val.empty()) {                              // a list or a nested map follows
                cur_list = key;
                lists[key];                                  // may stay empty if it is a map
                if (parent.empty()) { parent = key; parent_indent = indent; }
            } else {
                scalars[key] = val;
                cur_list.clear();
            }
        }
        auto need_s = [&](const char* k, std::string& out) {
            auto it = scalars.find(k);
            if (it == scalars.end()) { error = std::string("missing key: ") + k; return false; }
            out = it->second;
            return true;
        };
        auto need_l = [&](const char* k, std::vector<float>& out) {
            auto it = lists.find(k);
            if (it == lists.end() || it->second.empty()) {
                error = std::string("missing key: ") + k;
                return false;
            }
            out = it->second;
            return true;
        };
        std::string s;
        if (!need_s("env", env)) return false;
        if (!need_s("samples", s)) return false; samples = atoi(s.c_str());
        if (!need_s("state-dim", s)) return false; state_dim = atoi(s.c_str());
        if (!need_s("action-dim", s)) return false; act_dim = atoi(s.c_str());
        if (!need_s("horizon", s)) return false; horizon = atoi(s.c_str());
        if (!need_s("dt", s)) return false; dt = (float)atof(s.c_str());
        if (!need_s("lambda", s)) return false; lambda = (float)atof(s.c_str());
        if (!need_s("cost.type", cost_type)) return false;
        return need_l("noise", noise) && need_l("init-act", init_act) && need_l("max-a", max_a) &&
               need_l("goal", goal) && need_l("cost.w", cost_w);
    }

    bool parse_file(const std::string& path)
    {
        std::ifstream f(path);
        if (!f) { error = "cannot open " + path; return false; }
        std::stringstream ss;
        ss << f.rdbuf();
        return parse_text(ss.str());
    }

    // sizes the reference only warns about (src/main.cu:455-628) are errors here
    bool consistent()
    {
        if (state_dim != 2 * act_dim) { error = "state-dim must be 2*action-dim"; return false; }
        if ((int)goal.size() != state_dim || (int)cost_w.size() != state_dim ||
            (int)noise.size() != act_dim || (int)init_act.size() != act_dim ||
            (int)max_a.size() != act_dim) {
            error = "list sizes do not match the dimensions";
            return false;
        }
        return true;
    }
};

#endif  // MPPI_GPU_AMD_CONFIG_HPP_
