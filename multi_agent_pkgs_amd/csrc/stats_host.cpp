// stats_host.cpp — next row f3, ROS-free half: timing records and the shutdown report of Agent (see include/hdsm_stats.h).
// AC = multi_agent_planner/src/agent_class.cpp of the reference. Pure host C++.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <new>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/hdsm.h"
#include "../../include/hdsm_stats.h"

namespace {

struct Stats {
  int id = 0, n_rob = 1;
  std::vector<double> comp[6];
  std::vector<double> stamp;
  std::vector<std::vector<double>> state;
  std::vector<std::vector<double>> latency;  // [sender]
};

const char* kNames[6] = {"comp_time_sc_", "comp_time_tasc_", "comp_time_opt_", "comp_time_tot_", "comp_time_tot_wall_", "comp_time_path_"};

// The three figures the reference prints per timing series (SaveAndDisplayCompTime, AC:1943-1971): its running max starts at 0 and its
// running min at 1e10, which shows in the printout of an empty or an all-negative series, so the same clamps are applied here.
struct Summary {
  double mean, hi, lo;
  explicit Summary(const std::vector<double>& v)
      : mean(std::accumulate(v.begin(), v.end(), 0.0) / v.size()),
        hi(v.empty() ? 0.0 : std::max(0.0, *std::max_element(v.begin(), v.end()))),
        lo(v.empty() ? 1e10 : std::min(1e10, *std::min_element(v.begin(), v.end()))) {}
};

void write_csv_row(const std::vector<double>& v, const std::string& path) {
  std::ofstream f(path);
  f << std::fixed;
  for (double x : v) f << x << ",";
}

void comp_time(const std::vector<double>& v, const std::string& dir, const std::string& filename, bool save, std::ostream& out) {
  if (save) write_csv_row(v, dir + filename);
  const Summary m(v);
  out << filename << ": \nmean: " << m.mean << "\nmax: " << m.hi << "\nmin: " << m.lo << std::endl;
}

}  // namespace

extern "C" {

void* hdsm_stats_create(int32_t agent_id, int32_t n_rob) {
  if (n_rob < 1 || agent_id < 0 || agent_id >= n_rob) return nullptr;
  Stats* s = new (std::nothrow) Stats;
  if (!s) return nullptr;
  s->id = agent_id, s->n_rob = n_rob;
  s->latency.resize(n_rob);
  return s;
}

void hdsm_stats_destroy(void* stats) { delete static_cast<Stats*>(stats); }

int hdsm_stats_add(void* stats, int32_t kind, double ms) {
  Stats* s = static_cast<Stats*>(stats);
  if (!s || kind < 0 || kind > 5) return HDSM_ERR_BAD_ARG;
  s->comp[kind].push_back(ms);
  return HDSM_OK;
}

int hdsm_stats_add_state(void* stats, double stamp, const double* state, int32_t n_state) {
  Stats* s = static_cast<Stats*>(stats);
  if (!s || !state || n_state < 6) return HDSM_ERR_BAD_ARG;
  s->stamp.push_back(stamp);
  s->state.emplace_back(state, state + n_state);
  return HDSM_OK;
}

int hdsm_stats_add_latency(void* stats, int32_t from, double ms) {
  Stats* s = static_cast<Stats*>(stats);
  if (!s || from < 0 || from >= s->n_rob || from == s->id) return HDSM_ERR_BAD_ARG;
  s->latency[from].push_back(ms);
  return HDSM_OK;
}

int hdsm_stats_shutdown(void* stats, const char* dir_c, int32_t save_stats, char* report, int32_t report_cap) {
  Stats* s = static_cast<Stats*>(stats);
  if (!s || (save_stats && !dir_c)) return HDSM_ERR_BAD_ARG;
  std::string dir = dir_c ? dir_c : "";
  if (!dir.empty() && dir.back() != '/') dir += '/';
  const bool save = save_stats != 0;
  const std::string id = std::to_string(s->id);
  std::ostringstream out;
  for (int k = 0; k < 6; ++k) comp_time(s->comp[k], dir, std::string(kNames[k]) + id + ".csv", save, out);  // AC:2446-2459
  // SaveStateHistory, AC:1973-2008
  if (save) {
    std::ofstream f(dir + "state_hist_" + id + ".csv");
    for (size_t i = 0; i < s->state.size(); ++i) {
      f << std::fixed << s->stamp[i] << ",";
      for (size_t j = 0; j < s->state[i].size(); ++j) {
        f << std::fixed << s->state[i][j];
        if (j + 1 != s->state[i].size()) f << ",";
      }
      f << std::endl;
    }
  }
  double vel_average = 0, vel_max = 0;
  for (const auto& st : s->state) {
    const double vel = std::sqrt(st[3] * st[3] + st[4] * st[4] + st[5] * st[5]);
    vel_average += vel;
    if (vel > vel_max) vel_max = vel;
  }
  vel_average /= double(s->state.size());
  out << std::endl << "velocity for agent: " << s->id;
  out << std::endl << "mean: " << vel_average;
  out << std::endl << "max: " << vel_max << std::endl;
  // SaveAndDisplayCommunicationLatency, AC:2010-2061
  double total_mean = 0, total_max = 0;
  for (int i = 0; i < s->n_rob; ++i) {
    if (i == s->id) continue;
    double lat_mean = 0, lat_max = 0;
    for (double lat : s->latency[i]) {
      lat_mean += lat;
      if (lat > lat_max) lat_max = lat;
    }
    total_mean += lat_mean / double(s->latency[i].size());
    if (lat_max > total_max) total_max = lat_max;
  }
  total_mean = total_mean / (s->n_rob - 1);
  out << "communication latency (ms) for agent " << s->id << ": mean: " << total_mean << " max: " << total_max << std::endl;
  if (save) {
    std::ofstream f(dir + "com_latency_" + id + ".csv");
    for (int i = 0; i < s->n_rob; ++i) {
      if (i == s->id) continue;
      for (double lat : s->latency[i]) f << std::fixed << lat << ",";
      f << std::endl;
    }
  }
  const std::string text = out.str();
  if (report && report_cap > 0) {
    const size_t n = std::min((size_t)report_cap - 1, text.size());
    std::memcpy(report, text.data(), n);
    report[n] = 0;
  }
  return (int)text.size();
}

}  // extern "C"
