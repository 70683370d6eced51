// divergence_model.cpp -- ANALYSIS INFRASTRUCTURE (CPU only, never linked into the product): what would converged execution buy
// the headline kernel?  The kernel logic (librabft_simulator_amd/csrc/lbft_core.h) is compiled for the host with LBFT_HOST_SIG,
// which reports for every event of every network the set of LBFT_STAT points it passed (its "path signature").  A wavefront-step of
// the device executes the UNION of its lanes' paths, so with an instruction weight per path component (calibrated on the measured
// phase profile of DESIGN.md section 5: ~3 900 instructions per 32-lane wavefront-step) the model prices a step under different
// assignments of events to lanes:
//   unsorted        lane = network, as lbft_k_run0 runs today
//   kind-sorted     the events of G networks (G = 64, 128, 256) sorted by event kind before they are assigned to lanes
//   two-stage       ... and sorted again, after the pacemaker has decided, by what update_node will do (vote / propose / timeout /
//                   new QC / commit / messages to send); each stage boundary costs a state exchange through LDS
//   signature-sorted  sorted by the whole signature (not implementable -- the signature is only known afterwards --: the bound)
//   g++ -O2 -std=c++17 -Ioracle tests/tools/divergence_model.cpp -o /tmp/divergence_model -lpthread && /tmp/divergence_model [networks]
#define LBFT_HOST_SIG 1
#include <vector>
namespace lbft {
unsigned long long lbft_host_sig = 0;
static std::vector<unsigned long long>* g_steps = nullptr;
void lbft_host_step_done() { if (g_steps) g_steps->push_back(lbft_host_sig); lbft_host_sig = 0; }
}  // namespace lbft
#include "../../oracle/host_model.cpp"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64_;
static bool has(u64_ s, int k) { return (s >> k) & 1ULL; }

// Instruction weights of the path components (single-lane instruction counts, estimated from the flat phase profile of the 65 536 x 4
// batch: share of a wavefront-step x 3 900 instructions, every component being taken by some lane at nearly every step).
struct Cost { double stage1, stage2; int sends; };
static int msgs_of(u64_ s) { for (int m = 7; m >= 1; m--) if ((s >> (16 + m)) & 1ULL) return m; return 0; }
// coop_msgs >= 0: the wavefront's messages are sent lanes = messages (64 per pass, all lanes of the wavefront) instead of one loop
// iteration per message of the busiest lane; 120 instructions to redistribute them
static Cost price(u64_ u, int coop_msgs = -1) {  // u = union of the signatures of the lanes of one wavefront
  Cost c{0, 0, 0};
  c.stage1 += 300 + 220;                                  // queue pop, node-row burst (every event)
  if (has(u, 0)) c.stage1 += 40;                           // timer: duplicate bookkeeping, cancellation test
  if (has(u, 2)) c.stage1 += 100;                          // notification: snapshot words, epoch / certificate tests
  if (has(u, 32)) c.stage1 += 170;                         // ... first pending certificate (block fetch + insert_qc)
  if (has(u, 33)) c.stage1 += 170;                         // ... second
  if (has(u, 34)) c.stage1 += 125;                         // ... proposal (insert_block)
  if (has(u, 35)) c.stage1 += 60;                          // ... timeout certificate of the current round
  if (has(u, 36)) c.stage1 += 100;                         // ... timeouts
  if (has(u, 37)) c.stage1 += 75;                          // ... vote
  if (has(u, 3)) c.stage1 += 60;                           // request (answered by the requester: nothing but the response)
  if (has(u, 4)) c.stage1 += 20;                           // response
  if (has(u, 5)) c.stage2 += 160 + 60 + 80;                // update_node: pacemaker, tracker, node write-back
  if (has(u, 6)) c.stage2 += 90;                           // ... create_timeout
  if (has(u, 7)) c.stage2 += 120;                          // ... propose
  if (has(u, 8)) c.stage2 += 60;                           // ... a proposed block to consider
  if (has(u, 9)) c.stage2 += 110;                          // ... vote
  if (has(u, 10)) c.stage2 += 117;                         // ... new QC
  if (has(u, 11)) c.stage2 += has(u, 12) && !(u & (1ULL << 63)) ? 120 : 200;  // ... commits (bit 63: some lane took the slow path)
  if (has(u, 40)) c.stage2 += 30;                          // timer folded
  if (has(u, 41)) c.stage2 += 80;                          // timer scheduled
  for (int m = 1; m <= 7; m++) if (has(u, 16 + m)) c.sends = m;  // the send loop runs max(messages per lane) iterations
  if (has(u, 25) || has(u, 26)) c.stage2 += 100;           // notification snapshot written
  if (coop_msgs < 0) c.stage2 += 370.0 * c.sends;          // list preparation + delay sample + push, per iteration
  else if (coop_msgs > 0) c.stage2 += 95.0 * c.sends + 275.0 * ((coop_msgs + 63) / 64) + 120;  // (the shuffles stay per lane)
  return c;
}
static u64_ norm(u64_ s) {  // bit 63 = "commits, but not through the fast path"
  s &= ~(0xffULL << 48);    // (queue-length buckets)
  s &= ~((1ULL << 44) | (1ULL << 45) | (1ULL << 24));
  if (has(s, 11) && !has(s, 12)) s |= 1ULL << 63;
  return s;
}
static int kind_of(u64_ s) { return has(s, 2) ? 0 : has(s, 3) ? 1 : has(s, 4) ? 2 : has(s, 1) ? 4 : 3; }  // (cancelled timers: their own group)
static u64_ stage2_key(u64_ s) { return s & ((0xffULL << 5) | (0xffULL << 16) | (3ULL << 40) | (1ULL << 63) | (7ULL << 25)); }

int main(int argc, char** argv) {
  size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1024;
  lbft_oracle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_nodes = 4; cfg.delay_model = 0; cfg.mean = 10; cfg.variance = 4;
  cfg.commands_per_epoch = 30000; cfg.target_commit_interval = 100000; cfg.delta = 20; cfg.gamma = 2.0; cfg.lambda = 0.5;
  lbft_hostmodel_caps caps;
  memset(&caps, 0, sizeof(caps));
  caps.qcap = 4096; caps.scap = 64; caps.bcap = 512; caps.lcap = 512; caps.ql = 48;
  std::vector<std::vector<u64_>> sig(n);
  for (size_t i = 0; i < n; i++) {
    uint64_t seed = i + 1;
    lbft_oracle_counters c;
    uint32_t fault = 0, mq = 0, ms = 0;
    lbft::g_steps = &sig[i];
    lbft::lbft_host_sig = 0;
    int rc = lbft_hostmodel_run_batch(&cfg, &caps, &seed, 1, 1000, 1, nullptr, nullptr, nullptr, nullptr, 0, &c, &fault, &mq, &ms, nullptr, nullptr, nullptr, 0);
    if (rc) { fprintf(stderr, "instance %zu: rc %d fault %u\n", i, rc, fault); return 1; }
    for (auto& s : sig[i]) s = norm(s);
  }
  lbft::g_steps = nullptr;
  size_t max_steps = 0, total_events = 0;
  for (auto& v : sig) { max_steps = std::max(max_steps, v.size()); total_events += v.size(); }
  const int W = 32;
  const double XCHG = 150;  // a network's persistent state through LDS, in and out (per lane-step, all lanes active)
  struct Acc { double instr = 0; double wave_steps = 0; };
  auto run = [&](int G, int mode, bool coop = false) {  // mode 0 unsorted, 1 kind-sorted, 2 two-stage, 3 signature-sorted
    Acc a;
    std::vector<u64_> ev;
    for (size_t g0 = 0; g0 + G <= n; g0 += G) {
      for (size_t s = 0; s < max_steps; s++) {
        ev.clear();
        for (int l = 0; l < G; l++) if (s < sig[g0 + l].size()) ev.push_back(sig[g0 + l][s]);
        if (ev.empty()) break;
        if (mode == 1 || mode == 2) std::stable_sort(ev.begin(), ev.end(), [](u64_ x, u64_ y) { return kind_of(x) < kind_of(y); });
        if (mode == 3) std::sort(ev.begin(), ev.end());
        if (mode == 0) {  // lanes keep their networks: finished networks leave holes
          for (int w0 = 0; w0 < G; w0 += W) {
            u64_ u = 0; bool any = false; int tm = 0;
            for (int l = w0; l < w0 + W; l++) if (s < sig[g0 + l].size()) { u |= sig[g0 + l][s]; any = true; tm += msgs_of(sig[g0 + l][s]); }
            if (!any) continue;
            Cost c = price(u, coop ? tm : -1);
            a.instr += c.stage1 + c.stage2; a.wave_steps += 1;
          }
          continue;
        }
        size_t nw = (ev.size() + W - 1) / W;
        if (mode == 2) {
          std::vector<u64_> e2 = ev;
          std::stable_sort(e2.begin(), e2.end(), [](u64_ x, u64_ y) { return stage2_key(x) < stage2_key(y); });
          for (size_t w = 0; w < nw; w++) {
            u64_ u1 = 0, u2 = 0; int tm = 0;
            for (size_t l = w * W; l < std::min(ev.size(), (w + 1) * W); l++) { u1 |= ev[l]; u2 |= e2[l]; tm += msgs_of(e2[l]); }
            a.instr += price(u1).stage1 + price(u2, coop ? tm : -1).stage2 + 2 * XCHG; a.wave_steps += 1;
          }
        } else {
          for (size_t w = 0; w < nw; w++) {
            u64_ u = 0;
            for (size_t l = w * W; l < std::min(ev.size(), (w + 1) * W); l++) u |= ev[l];
            Cost c = price(u);
            a.instr += c.stage1 + c.stage2 + XCHG; a.wave_steps += 1;
          }
        }
      }
    }
    return a;
  };
  double single = 0;
  for (auto& v : sig) for (u64_ s : v) { Cost c = price(s); single += c.stage1 + c.stage2; }
  printf("%zu networks x 4 nodes, max_clock 1000: %.1f events (queue pops) per network; single-path instructions per event %.0f\n", n, (double)total_events / n,
         single / total_events);
  Acc base = run(32, 0);
  printf("%-44s %9.0f instr / wavefront-step   (%.2f x the single path; measured: 3 900, 4.7 x)\n", "unsorted, lane = network (lbft_k_run0)", base.instr / base.wave_steps,
         base.instr / base.wave_steps / (single / total_events));
  const char* names[4] = {"", "kind-sorted", "two-stage (kind, then update_node's actions)", "signature-sorted (bound)"};
  for (int mode = 1; mode <= 3; mode++)
    for (int G : {64, 128, 256}) {
      Acc a = run(G, mode);
      char label[96];
      snprintf(label, sizeof label, "%s, %d networks", names[mode], G);
      printf("%-52s %9.0f instr / wavefront-step, total %.3f of unsorted\n", label, a.instr / a.wave_steps, a.instr / base.instr);
    }
  Acc cs = run(32, 0, true);
  printf("%-52s %9.0f instr / wavefront-step, total %.3f of unsorted\n", "unsorted + cooperative sends (lanes = messages)", cs.instr / cs.wave_steps, cs.instr / base.instr);
  for (int G : {64, 128, 256}) {
    Acc a = run(G, 2, true);
    char label[96];
    snprintf(label, sizeof label, "two-stage + cooperative sends, %d networks", G);
    printf("%-52s %9.0f instr / wavefront-step, total %.3f of unsorted\n", label, a.instr / a.wave_steps, a.instr / base.instr);
  }
  return 0;
}
