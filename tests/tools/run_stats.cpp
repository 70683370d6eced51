// run_stats.cpp -- TEST / ANALYSIS INFRASTRUCTURE (CPU only): how much of a large network's event stream the cooperative RUNS of the event loop take
// (coop_requests / coop_responses, lbft_core.h), counted by the LBFT_STAT points in a host build of the kernel logic (oracle/host_model.cpp, the
// cooperative loop with 64 emulated lanes).
//   g++ -O2 -std=c++17 -Ioracle tests/tools/run_stats.cpp -o /tmp/run_stats -lpthread
//   /tmp/run_stats nodes instances max_clock variance equivocate_every quirks commands_per_epoch rights_rotation weighted
#define LBFT_HOST_STATS 1
namespace lbft { unsigned long long lbft_host_stats[64]; }
#include "../../oracle/host_model.cpp"
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
  auto arg = [&](int i, double d) { return argc > i ? atof(argv[i]) : d; };
  lbft_oracle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_nodes = (uint32_t)arg(1, 64);
  size_t n = (size_t)arg(2, 2);
  int64_t max_clock = (int64_t)arg(3, 300);
  cfg.delay_model = 0; cfg.mean = 10; cfg.variance = arg(4, 4.0);
  cfg.equivocate_every = (uint32_t)arg(5, 0);
  cfg.quirks = (uint32_t)arg(6, 0);
  cfg.commands_per_epoch = (uint64_t)arg(7, 30000);
  cfg.rights_rotation = (uint32_t)arg(8, 0);
  std::vector<uint64_t> rights(cfg.num_nodes);
  if (arg(9, 0) != 0) { for (uint32_t i = 0; i < cfg.num_nodes; i++) rights[i] = 1 + (i % 4); cfg.voting_rights = rights.data(); }
  cfg.target_commit_interval = 100000; cfg.delta = 20; cfg.gamma = 2.0; cfg.lambda = 0.5; cfg.math_mode = 1;
  lbft_hostmodel_caps caps;
  memset(&caps, 0, sizeof(caps));
  const uint32_t nn = cfg.num_nodes;
  caps.qcap = 8 * nn * nn; caps.scap = (cfg.quirks & 1) ? nn * nn + 8 * nn : 8 * nn; caps.bcap = (uint32_t)(max_clock / 10 + 64); caps.lcap = caps.bcap; caps.qheap = 1; caps.tw = 1;
  caps.qcal = 1; caps.ring = 256;
  std::vector<uint64_t> seeds(n);
  for (size_t i = 0; i < n; i++) seeds[i] = i + 1;
  lbft_oracle_counters c;
  std::vector<uint32_t> maxq(n), maxsnap(n), faults(n);
  int rc = lbft_hostmodel_run_batch(&cfg, &caps, seeds.data(), n, max_clock, 1, nullptr, nullptr, nullptr, nullptr, 0, &c, faults.data(), maxq.data(),
                                    maxsnap.data(), nullptr, nullptr, nullptr, 0);
  const unsigned long long* s = lbft::lbft_host_stats;
  double ev = (double)(c.events[0] + c.events[1] + c.events[2] + c.events[3]);
  printf("{\"rc\": %d, \"nodes\": %u, \"instances\": %zu, \"max_clock\": %lld, \"quirks\": %u, \"events\": {\"notify\": %llu, \"request\": %llu, \"response\": %llu, \"timer\": %llu},\n"
         " \"request_runs\": {\"runs\": %llu, \"events\": %llu, \"share_of_requests\": %.4f, \"events_per_run\": %.2f},\n"
         " \"response_runs\": {\"runs\": %llu, \"events\": %llu, \"share_of_responses\": %.4f, \"events_per_run\": %.2f, \"ended_by_an_update_that_does_something\": %llu, \"consumed_nothing\": %llu},\n"
         " \"notification_runs\": {\"runs\": %llu, \"events\": %llu, \"share_of_notifications\": %.4f, \"events_per_run\": %.2f, \"consumed_nothing\": %llu},\n"
         " \"responses_under_quirks_bit_0\": {\"nothing_to_insert\": %llu, \"and_update_is_a_no_op\": %llu, \"other\": %llu, \"no_op_and_timer_folded_in_streaks_of_2_or_more\": %llu, \"such_streaks\": %llu, \"in_streaks_of_4_or_more\": %llu},\n"
         " \"notifications\": {\"ordinary_steps\": %llu, \"leaving_the_node_as_it_was\": %llu, \"of_them_in_streaks_of_4_or_more\": %llu, \"of_8_or_more\": %llu},\n"
         " \"share_of_all_events_in_runs\": %.4f, \"ordinary_steps\": {\"notify\": %llu, \"request\": %llu, \"response\": %llu, \"timer_pops\": %llu}}\n",
         rc, cfg.num_nodes, n, (long long)max_clock, cfg.quirks, (unsigned long long)c.events[0], (unsigned long long)c.events[1], (unsigned long long)c.events[2],
         (unsigned long long)c.events[3], s[30], s[31], c.events[1] ? (double)s[31] / c.events[1] : 0.0, s[30] ? (double)s[31] / s[30] : 0.0,
         s[13], s[14], c.events[2] ? (double)s[14] / c.events[2] : 0.0, s[13] ? (double)s[14] / s[13] : 0.0, s[15], s[29],
         s[60], s[61], c.events[0] ? (double)s[61] / c.events[0] : 0.0, s[60] ? (double)s[61] / s[60] : 0.0, s[62],
         s[33], s[38], s[39], s[56], s[57], s[63], s[43], s[47], s[58], s[59], ev ? (double)(s[31] + s[14] + s[61]) / ev : 0.0, s[2], s[3], s[4], s[0]);
  return rc;
}
