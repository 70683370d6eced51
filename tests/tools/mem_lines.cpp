// mem_lines.cpp -- TEST / ANALYSIS INFRASTRUCTURE (CPU only; round-5 review item 4): which structure of an instance's HBM rows costs how many 128-byte
// lines per processed event?  A host build of the kernel logic (oracle/host_model.cpp with the LBFT_HOST_MEMHOOK / LBFT_HOST_BUCKETS hooks of lbft_core.h)
// runs a large-network configuration exactly as the device does (class 2, calendar queue, ring of draws, cooperative bulk sends with 64 emulated lanes,
// the 32-entry LDS window of block records) and records every access to the rows.  Per event the DISTINCT lines read and the distinct lines written are
// counted per structure (a line touched twice by one event is one line: it is still in the L2; between two events of the same network thousands of
// other networks run, so nothing is assumed to survive).  Reconciles with rocprofv3's FETCH_SIZE / WRITE_SIZE of the same configuration
// (profiles/r05/*.pmc.json): lines_read x 128 B vs 2 x FETCH_SIZE, lines_written x 64 B vs WRITE_SIZE.
//   g++ -O2 -std=c++17 -Ioracle tests/tools/mem_lines.cpp -o /tmp/mem_lines -lpthread && /tmp/mem_lines <c4|c4live|c5|c5live|c5named> [instances]
#define LBFT_HOST_BUCKETS 1
#define LBFT_HOST_MEMHOOK 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../oracle/host_model.cpp"

enum Region { R_SCALARS, R_NODE_FIXED, R_NODE_HCBR, R_NODE_SETS, R_Q_NEXT, R_Q_FREE, R_Q_META, R_CAL_HEAD, R_CAL_TAIL, R_CAL_BM, R_SNAP_FIXED, R_SNAP_HCBR, R_SNAP_EXT,
              R_SNAP_REF, R_SNAP_FREE, R_BLK_HOT, R_BLK_COLD, R_BLK_EXT, R_LOG, R_LIST, R_ARCH, R_RARCH, R_SYNC, R_RING, R_N };
static const char* RN[R_N] = {"instance scalars", "node: fixed words", "node: hcbr buffers", "node: set extension words", "queue: key / link rows (heap only)", "queue: stack of freed chunks",
                              "queue: chunk pool (event metas + links)", "calendar: (head, tail) pairs", "calendar: tails (unused)", "calendar: bitmap", "snapshot: fixed words", "snapshot: hcbr words",
                              "snapshot: set ext + request", "snapshot: reference counts", "snapshot: free stack", "block: hot record", "block: time / command / voters",
                              "block: set extension words", "commit logs", "receiver list", "epoch archive (record exchange)", "retired stores", "sync scratch list", "ring of draws"};
static lbft::Params g_p;
static bool g_have_p = false;
static Region classify(unsigned w) {
  const lbft::Params& p = g_p;
  using namespace lbft;
  if (w < p.off_node) return R_SCALARS;
  if (w < p.off_qhi) { unsigned f = (w - p.off_node) % p.node_words; return f < NF_FIXED_WORDS ? R_NODE_FIXED : f < node_hcbr_off(p.mw) ? R_NODE_SETS : R_NODE_HCBR; }
  if (w < p.off_qlo) return R_Q_NEXT;     // (heap queue only: the calendar keeps no key / link rows since round 6)
  if (w < p.off_qmeta) return R_Q_FREE;   // calendar: stack of freed chunks
  if (w < p.off_cal_head) return R_Q_META;  // calendar: chunk pool (31 event metas + a link word per 128-byte chunk)
  if (w < p.off_cal_bm) return R_CAL_HEAD;  // (head, tail) word pairs
  if (w < p.off_snap) return R_CAL_BM;
  if (w < p.off_snap_ref) { unsigned f = (w - p.off_snap) % p.snap_words; return f < S_FIXED_WORDS ? R_SNAP_FIXED : f < S_FIXED_WORDS + 2 * p.n ? R_SNAP_HCBR : R_SNAP_EXT; }
  if (w < p.off_snap_free) return R_SNAP_REF;
  if (w < p.off_blk) return R_SNAP_FREE;
  if (w < p.off_log) { unsigned f = (w - p.off_blk) % p.blk_words; return f < BC_WORDS ? R_BLK_HOT : f < B_WORDS ? R_BLK_COLD : R_BLK_EXT; }
  if (w < p.off_list) return R_LOG;
  if (w < p.off_trace) return R_LIST;
  if (w < p.off_arch) return R_LIST;
  if (w < p.off_rarch) return R_ARCH;
  if (w < p.off_sync) return R_RARCH;
  if (w < p.off_ring) return R_SYNC;
  return R_RING;
}

static std::unordered_set<unsigned> g_rd, g_wr;  // lines of the current event (line = byte offset / 128, region in the high bits is not needed: regions do not share lines... they may at borders: counted once for the first region)
static unsigned long long g_lr[R_N], g_lw[R_N], g_ar[R_N], g_aw[R_N], g_events, g_kind_events[4], g_kind_lr[4], g_kind_lw[4];
static int g_cur_kind = -1;
static std::vector<std::pair<unsigned, int>> g_acc;  // (byte offset, store) of the current event
static void flush() {
  if (g_cur_kind < 0) { g_acc.clear(); return; }
  g_rd.clear(); g_wr.clear();
  unsigned long long lr = 0, lw = 0;
  for (auto& a : g_acc) {
    unsigned line = a.first >> 7;
    Region r = classify(a.first >> 2);
    if (a.second) { g_aw[r]++; if (g_wr.insert(line).second) { g_lw[r]++; lw++; } }
    else { g_ar[r]++; if (!g_wr.count(line) && g_rd.insert(line).second) { g_lr[r]++; lr++; } }  // (a line the event wrote first is in the L2 already)
  }
  g_events++; g_kind_events[g_cur_kind]++; g_kind_lr[g_cur_kind] += lr; g_kind_lw[g_cur_kind] += lw;
  g_acc.clear();
}
namespace lbft {
void lbft_host_pop(int, unsigned kind, unsigned, unsigned) { flush(); g_cur_kind = (int)kind; }
void lbft_host_push(long long, unsigned, unsigned) {}
void lbft_host_mem(unsigned off, int store) { g_acc.push_back({off, store}); }
}  // namespace lbft

int main(int argc, char** argv) {
  std::string name = argc > 1 ? argv[1] : "c5";
  size_t n_inst = argc > 2 ? (size_t)atol(argv[2]) : 1;
  lbft_oracle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.delay_model = 0; cfg.mean = 10; cfg.variance = 4; cfg.commands_per_epoch = 30000;
  cfg.target_commit_interval = 100000; cfg.delta = 20; cfg.gamma = 2.0; cfg.lambda = 0.5;
  int64_t max_clock = 300;
  std::vector<uint64_t> rights;
  auto weighted = [&]() { rights.resize(100); for (int i = 0; i < 100; i++) rights[i] = 1 + (i % 4); cfg.voting_rights = rights.data(); };
  if (name == "c4") { cfg.num_nodes = 64; cfg.variance = 400; cfg.equivocate_every = 5; }
  else if (name == "c4live") { cfg.num_nodes = 64; cfg.variance = 400; cfg.equivocate_every = 5; cfg.quirks = 3; max_clock = 1000; }
  else if (name == "c5") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 50; }
  else if (name == "c5live") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 3; cfg.quirks = 3; cfg.rights_rotation = 1; max_clock = 450; }
  else if (name == "c5named") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 50; cfg.quirks = 3; cfg.rights_rotation = 1; max_clock = 2500; }
  else { fprintf(stderr, "config: c4 | c4live | c5 | c5live | c5named\n"); return 2; }
  const unsigned n = cfg.num_nodes;
  lbft_hostmodel_caps caps;
  memset(&caps, 0, sizeof(caps));
  // capacities as lbft_hip.hip prepare_run computes them; the device's execution mode for large networks
  caps.qcap = 8 * n * n; caps.scap = (cfg.quirks & 1) ? n * n + 8 * n : 8 * n; caps.bcap = (uint32_t)(max_clock / 10 + 64); caps.lcap = caps.bcap; caps.qheap = 1; caps.tw = 1;
  caps.qcal = 1; caps.ring = 512; caps.ring_topup = 4;
  {  // the layout the run will use (same computation as lbft_hostmodel_run_batch)
    using namespace lbft;
    Params& p = g_p;
    memset(&p, 0, sizeof(p));
    p.n = n; p.m = 1; p.stride = 64; p.tw = 1; p.rsh = 2; p.qcap = caps.qcap; p.scap = caps.scap; p.bcap = caps.bcap; p.lcap = caps.lcap; p.max_clock = (i32)max_clock;
    p.qheap = 1; p.qcal = 1; p.quirks = cfg.quirks; p.equiv = cfg.equivocate_every; p.ring = 512;
    u64 eauto = (u64)caps.bcap / cfg.commands_per_epoch + 2;
    p.ecap = (u32)(eauto > 4096 ? 4096 : eauto); if (p.ecap < 64) p.ecap = 64;
    compute_layout(p);
    g_have_p = true;
  }
  unsigned long long fold = 0;
  for (size_t inst = 0; inst < n_inst; inst++) {
    uint64_t seed = inst + 1;
    lbft_oracle_counters c;
    uint32_t fault = 0, mq = 0, ms = 0;
    g_cur_kind = -1; g_acc.clear();
    int rc = lbft_hostmodel_run_batch(&cfg, &caps, &seed, 1, max_clock, 1, nullptr, nullptr, nullptr, nullptr, 0, &c, &fault, &mq, &ms, nullptr, nullptr, nullptr, 0);
    g_acc.clear(); g_cur_kind = -1;  // (the read-back after the run is not part of the event loop; the last event's accesses are dropped with it)
    if (rc != 0 || fault) { fprintf(stderr, "instance %zu: rc %d fault %x\n", inst, rc, fault); return 1; }
    fold += c.events[0] + c.events[1] + c.events[2] + c.events[3];
  }
  unsigned long long LR = 0, LW = 0;
  for (int r = 0; r < R_N; r++) { LR += g_lr[r]; LW += g_lw[r]; }
  printf("{\"config\": \"%s\", \"instances\": %zu, \"bytes_per_instance\": %llu, \"queue_pops\": %llu, \"reference_equivalent_events\": %llu,\n", name.c_str(), n_inst,
         (unsigned long long)g_p.total_words * 4ULL, g_events, fold);
  printf(" \"lines_read_per_pop\": %.3f, \"lines_written_per_pop\": %.3f, \"model_bytes_per_pop\": {\"fetch_128B_lines\": %.0f, \"write_64B\": %.0f},\n", (double)LR / g_events,
         (double)LW / g_events, 128.0 * LR / g_events, 64.0 * LW / g_events);
  printf(" \"by_event_kind\": {");
  static const char* KN[4] = {"notify", "request", "response", "timer"};
  for (int k = 0; k < 4; k++)
    printf("\"%s\": {\"share\": %.4f, \"lines_read\": %.2f, \"lines_written\": %.2f}%s", KN[k], (double)g_kind_events[k] / g_events,
           g_kind_events[k] ? (double)g_kind_lr[k] / g_kind_events[k] : 0.0, g_kind_events[k] ? (double)g_kind_lw[k] / g_kind_events[k] : 0.0, k < 3 ? ", " : "},\n");
  printf(" \"by_structure\": [\n");
  std::vector<int> order(R_N);
  for (int r = 0; r < R_N; r++) order[r] = r;
  std::sort(order.begin(), order.end(), [](int a, int b) { return g_lr[a] * 2 + g_lw[a] > g_lr[b] * 2 + g_lw[b]; });
  bool first = true;
  for (int r : order) {
    if (!g_lr[r] && !g_lw[r]) continue;
    printf("%s  {\"structure\": \"%s\", \"lines_read_per_pop\": %.3f, \"lines_written_per_pop\": %.3f, \"loads_per_pop\": %.2f, \"stores_per_pop\": %.2f}", first ? "" : ",\n", RN[r],
           (double)g_lr[r] / g_events, (double)g_lw[r] / g_events, (double)g_ar[r] / g_events, (double)g_aw[r] / g_events);
    first = false;
  }
  printf("\n ]}\n");
  return 0;
}
