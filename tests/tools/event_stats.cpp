// event_stats.cpp -- TEST / ANALYSIS INFRASTRUCTURE (CPU only): per-instance statistics of the event loop, counted by the
// LBFT_STAT points of librabft_simulator_amd/csrc/lbft_core.h in a host build of the kernel logic (oracle/host_model.cpp).
// These numbers decide which paths of a wavefront-step are "always taken by some lane" (DESIGN.md section 5):
//   g++ -O2 -std=c++17 -Ioracle tests/tools/event_stats.cpp -o /tmp/event_stats -lpthread && /tmp/event_stats [nodes] [instances] [quirks] [commands_per_epoch] [rights_rotation]
#define LBFT_HOST_STATS 1
namespace lbft { unsigned long long lbft_host_stats[64]; }
#include "../../oracle/host_model.cpp"
#include <cstdio>
#include <cstdlib>

static const char* kNames[64] = {
    /* 0*/ "timer events popped", "... cancelled (ignore_scheduled_updates_until)", "notify events", "request events", "response events",
    /* 5*/ "update_node calls", "... create_timeout", "... propose", "... has a proposed block to consider", "... votes",
    /*10*/ "... forms a QC", "process_commits with something to commit", "... through the no-row-read fast path", nullptr, nullptr, nullptr,
    /*16*/ "events sending 0 messages", "1 message", "2 messages", "3 messages", "4 messages", "5 messages", "6 messages", ">= 7 messages",
    /*24*/ "messages sent", "broadcasts", "single sends", "query_all", "sync requests", nullptr, nullptr, nullptr,
    /*32*/ "notification: first pending certificate", "second pending certificate", "carries a proposal", "TC of the current round",
    /*36*/ "timeouts of the current round", "carries a vote", nullptr, nullptr,
    /*40*/ "timers folded into a pending one", "timers scheduled", "... beyond max_clock", nullptr,
    /*44*/ "block-record lookups", "... missing the register cache", "responses going on with a later epoch at the next step (quirks bit 0)", nullptr,
    /*48*/ "pops at queue length 0-7", "8-15", "16-23", "24-31", "32-39", "40-47", "48-55", ">= 56"};

int main(int argc, char** argv) {
  lbft_oracle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_nodes = argc > 1 ? atoi(argv[1]) : 4;
  cfg.delay_model = 0; cfg.mean = 10; cfg.variance = 4;
  cfg.commands_per_epoch = argc > 4 ? atoi(argv[4]) : 30000; cfg.quirks = argc > 3 ? atoi(argv[3]) : 0; cfg.rights_rotation = argc > 5 ? atoi(argv[5]) : 0;
  cfg.target_commit_interval = 100000; cfg.delta = 20; cfg.gamma = 2.0; cfg.lambda = 0.5;
  lbft_hostmodel_caps caps;
  memset(&caps, 0, sizeof(caps));
  caps.qcap = 4096; caps.scap = cfg.quirks & 1 ? 512 : 64; caps.bcap = 512; caps.lcap = 512; caps.ql = cfg.quirks & 1 ? 0 : 48;
  if (cfg.quirks & 1) caps.qheap = 1;
  size_t n = argc > 2 ? (size_t)atol(argv[2]) : 512;
  std::vector<uint64_t> seeds(n);
  for (size_t i = 0; i < n; i++) seeds[i] = i + 1;
  lbft_oracle_counters c;
  std::vector<uint32_t> maxq(n), maxsnap(n), faults(n);
  int rc = lbft_hostmodel_run_batch(&cfg, &caps, seeds.data(), n, 1000, 1, nullptr, nullptr, nullptr, nullptr, 0, &c, faults.data(), maxq.data(),
                                    maxsnap.data(), nullptr, nullptr, nullptr, 0);
  printf("rc=%d  %zu instances x %u nodes, LogNormal(10,4), max_clock 1000\n", rc, n, cfg.num_nodes);
  printf("events per instance: notify %.1f request %.1f response %.1f timer %.1f (timers include folded duplicates); scheduled %.1f; rounds %.1f\n",
         (double)c.events[0] / n, (double)c.events[1] / n, (double)c.events[2] / n, (double)c.events[3] / n, (double)c.events_scheduled / n,
         (double)c.rounds / n);
  for (int k = 0; k < 64; k++)
    if (lbft::lbft_host_stats[k]) printf("  %10.3f  %s\n", (double)lbft::lbft_host_stats[k] / n, kNames[k] ? kNames[k] : "?");
  return rc;
}
