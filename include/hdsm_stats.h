/*
 * hdsm_stats.h — next row f3, the ROS-free half: the per-stage timing records of multi_agent_planner::Agent and its
 * shutdown report (AC = multi_agent_planner/src/agent_class.cpp of lis-epfl/multi_agent_pkgs).
 *
 * The reference appends one value per replan to comp_time_sc_ / comp_time_tasc_ / comp_time_opt_ / comp_time_tot_ /
 * comp_time_tot_wall_ (AC:1446, 1214, 1022, 193, 198-206) and one per path update to comp_time_path_ (AC:432), the
 * planned state to state_hist_ (AC:240-245), one latency per received trajectory to com_latency_ms_[sender] (AC:637-642);
 * Agent::OnShutdown (AC:2446-2466) writes them as CSV files (when save_stats) and prints mean / max / min.
 * These functions keep the same records and write the SAME files in the same format:
 *   comp_time_{sc,tasc,opt,tot,tot_wall,path}_<id>.csv   one line, every value as std::fixed (6 decimals) followed by ","
 *                                                       (SaveAndDisplayCompTime, AC:1943-1971)
 *   state_hist_<id>.csv                                 one line per record: stamp,then the state, comma separated
 *                                                       (SaveStateHistory, AC:1973-2008)
 *   com_latency_<id>.csv                                one line per OTHER agent: its latencies, each followed by ","
 *                                                       (SaveAndDisplayCommunicationLatency, AC:2010-2061)
 * and return the text the reference prints to std::cout.
 * With the batched solver one launch serves every local agent: hdsm_swarm_* (hdsm_swarm.h) records the corridor time per
 * agent, the caller hands in the duration of the fused launch (tasc + opt of every agent: recorded as opt, tasc = 0).
 */
#ifndef HDSM_STATS_H
#define HDSM_STATS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum hdsm_stat_kind {
  HDSM_STAT_SC = 0, HDSM_STAT_TASC = 1, HDSM_STAT_OPT = 2, HDSM_STAT_TOT = 3, HDSM_STAT_TOT_WALL = 4, HDSM_STAT_PATH = 5
} hdsm_stat_kind;

void* hdsm_stats_create(int32_t agent_id, int32_t n_rob);
void hdsm_stats_destroy(void* stats);
int hdsm_stats_add(void* stats, int32_t kind, double milliseconds);
int hdsm_stats_add_state(void* stats, double stamp, const double* state, int32_t n_state);
int hdsm_stats_add_latency(void* stats, int32_t from_agent, double milliseconds);
/* Agent::OnShutdown: writes the CSV files into `dir` when save_stats != 0 and returns, in report[report_cap], the text the
 * reference prints (truncated if it does not fit); the return value is the length of the full text or a negative hdsm_error. */
int hdsm_stats_shutdown(void* stats, const char* dir, int32_t save_stats, char* report, int32_t report_cap);

#ifdef __cplusplus
}
#endif
#endif
