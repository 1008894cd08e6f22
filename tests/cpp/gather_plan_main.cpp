// CPU test driver (no GPU, no RCCL) of the gather's bookkeeping, fm-radio_amd/csrc/fmd_gather_plan.h — the part of libfmdgather.so that
// decides who collects block k, where its shards land and when a buffer set may be reused (reference anchor: one demodulator wired to
// an audio observer and an RDS byte chain per station, src/app.cpp:19-34; here n_ranks shards of them).
//   gather_plan_main plan      configurations accepted / refused, collectors, copies against RCCL shards, receive lists
//   gather_plan_main run R K   R rank threads and a collecting thread run K blocks through the hand-shake with simulated buffers:
//                              every block arrives complete, on the collector the plan names, and no buffer set is overwritten while the
//                              collector still holds its views (rotation and root-only)
//   gather_plan_main abort     a rank that stops submitting: abort() releases the collector and every other rank within a timeout
// Prints one JSON line; exit code 0 only if every check held.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fmd_gather_plan.h"

using namespace fmd_gather_plan;

static int fails = 0;
#define CHECK(x) do { if (!(x)) { fprintf(stderr, "check failed: %s (line %d)\n", #x, __LINE__); fails++; } } while (0)

static int test_plan() {
    {   // one rank per device, rotating
        const int dev[4] = {0, 1, 2, 3};
        Plan p; CHECK(p.init(4, dev, 1, kRotate).empty());
        CHECK(p.rotate && p.collectors() == 4);
        for (long k = 0; k < 12; k++) {
            CHECK(p.collector_rank(k) == (1 + k) % 4);
            CHECK(p.collector_device(k) == dev[(1 + k) % 4]);
            CHECK(p.collector_index(k) == k % 4 && Plan::slot(k) == k % kDepth);
            const auto rv = p.receives(k);
            CHECK(rv.size() == 3);                                   // everybody but the collector itself crosses RCCL
            for (int r = 0; r < 4; r++) CHECK(p.via_rccl(r, k) == (r != p.collector_rank(k)));
            const auto po = p.polled(k);
            CHECK(po.size() == 1 && po[0] == p.collector_rank(k));   // its receives cover the rest
        }
        for (int r = 0; r < 4; r++) CHECK(p.needs_pcm_staging(r));
    }
    {   // every rank on ONE device (a one-GPU box), rotating: every hand-over is a copy, every collector lives on that device
        const int dev[3] = {0, 0, 0};
        Plan p; CHECK(p.init(3, dev, 0, kRotate).empty());
        CHECK(p.rotate && p.collectors() == 3 && p.uniq.size() == 1);
        for (long k = 0; k < 9; k++) {
            CHECK(p.collector_rank(k) == k % 3 && p.collector_device(k) == 0);
            CHECK(p.receives(k).empty());
            CHECK(p.polled(k).size() == 3);
            for (int r = 0; r < 3; r++) CHECK(!p.via_rccl(r, k));
        }
        for (int r = 0; r < 3; r++) CHECK(!p.needs_pcm_staging(r));
    }
    {   // root only; two ranks share the root's device, the third has its own
        const int dev[3] = {0, 0, 1};
        Plan p; CHECK(p.init(3, dev, 0, 0).empty());
        CHECK(!p.rotate && p.collectors() == 1);
        for (long k = 0; k < 5; k++) { CHECK(p.collector_rank(k) == 0); CHECK(p.receives(k) == std::vector<int>{2}); CHECK((p.polled(k) == std::vector<int>{0, 1})); }
        CHECK(p.comm_index[0] == 0 && p.comm_index[1] == 0 && p.comm_index[2] == 1);
        CHECK(!p.needs_pcm_staging(0) && !p.needs_pcm_staging(1) && p.needs_pcm_staging(2));
    }
    {   // refused configurations
        const int shared_remote[3] = {0, 1, 1};
        Plan p; CHECK(!p.init(3, shared_remote, 0, 0).empty());                                   // two ranks on a device that is not the collector's
        const int mixed[3] = {0, 0, 1};
        Plan q; CHECK(!q.init(3, mixed, 0, kRotate).empty());                                     // rotation over a mixed placement
        const int two[2] = {0, 0};
        Plan r; CHECK(!r.init(2, two, 0, kLoopbackRccl).empty());                                 // loop-back needs the collector alone on its device
        const int one[1] = {0};
        Plan s; CHECK(s.init(1, one, 0, kLoopbackRccl).empty() && s.via_rccl(0, 0) && s.receives(0) == std::vector<int>{0});
        Plan t; CHECK(!t.init(1, one, 1, 0).empty() && !t.init(0, one, 0, 0).empty() && !t.init(1, one, 0, 64).empty());
        Plan u; CHECK(u.init(1, one, 0, kRotate).empty() && !u.rotate);                           // one rank: nothing to rotate
    }
    {   // a handle's block numbering against the gather's (epoch = fmd_outputs_epoch: changes only with fmd_reset)
        BlockBase b; b.start(-1, 1);                  // fresh handle
        CHECK(b.relative(0, 1, 0) == 0 && b.relative(1, 1, 1) == 1);
        BlockBase c; c.start(4, 1);                   // five blocks of pre-roll before the gather existed
        CHECK(c.relative(5, 1, 0) == 0 && c.relative(6, 1, 1) == 1);
        CHECK(c.relative(0, 2, 2) == 2 && c.relative(1, 2, 3) == 3);      // fmd_reset in between: the numbering restarted, the first block re-bases
        BlockBase e; e.start(-1, 1);
        CHECK(e.relative(0, 1, 0) == 0 && e.relative(2, 1, 1) == 2);      // a skipped block is NOT the gather's block 1: reported to the caller
        // a second submit without a new block must NOT pass as a restart (it did while blk <= last meant "reset")
        BlockBase f; f.start(-1, 1);
        CHECK(f.relative(0, 1, 0) == 0);
        CHECK(f.relative(0, 1, 1) != 1);              // the same block again: refused
        CHECK(f.relative(1, 1, 1) == 1);              // ... and the gather carries on with the right one
        // a handle under fmd_set_output_lag shows no block, then the block before: neither is the gather's block
        BlockBase l; l.start(-1, 1);
        CHECK(l.relative(-1, 1, 0) != 0);
        CHECK(l.relative(0, 1, 1) != 1);
        // a reset followed by TWO blocks before the next submit: a skipped block, not a restart at the second one
        BlockBase r; r.start(-1, 1);
        CHECK(r.relative(0, 1, 0) == 0 && r.relative(1, 2, 1) != 1);
        // a reset before the gather's first submit
        BlockBase q; q.start(7, 3);
        CHECK(q.relative(0, 4, 0) == 0 && q.relative(1, 4, 1) == 1);
    }
    return fails;
}

// simulated device side: a shard "arrives" when its rank writes its tag into the collector's buffer set
struct Sim {
    Plan plan; Sync sync; int R; long K; bool rotate;
    std::vector<std::vector<std::vector<long>>> buf;      // [collector index][slot][rank] = block whose shard lies there
    std::atomic<long> holding{-1};                        // block whose views the collector holds right now
    std::atomic<int> overwrites{0};
    Sim(int R_, long K_, bool rot, const std::vector<int>& dev) : sync(R_), R(R_), K(K_), rotate(rot) {
        const std::string why = plan.init(R_, dev.data(), 0, rot ? kRotate : 0u);
        if (!why.empty()) { fprintf(stderr, "%s\n", why.c_str()); fails++; }
        buf.assign((size_t)plan.collectors(), std::vector<std::vector<long>>(kDepth, std::vector<long>((size_t)R_, -1)));
    }
};

static int test_run(int R, long K) {
    for (int rot = 0; rot < 2; rot++) {
        std::vector<int> dev((size_t)R);
        for (int r = 0; r < R; r++) dev[(size_t)r] = rot ? r : 0;      // rotation: one device per rank; root only: everybody on the root's device
        Sim s(R, K, rot != 0, dev);
        std::vector<std::thread> th;
        for (int r = 0; r < R; r++) th.emplace_back([&s, r] {
            for (long k = 0; k < s.K; k++) {
                if (!s.sync.begin_submit(k)) return;
                std::vector<long>& set = s.buf[(size_t)s.plan.collector_index(k)][(size_t)Plan::slot(k)];
                // the set must not be the one whose views the collector still holds
                const long h = s.holding.load();
                if (h >= 0 && s.plan.collector_index(h) == s.plan.collector_index(k) && Plan::slot(h) == Plan::slot(k) && h != k) s.overwrites++;
                set[(size_t)r] = k;
                if ((k + r) % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
                s.sync.end_submit(r, k);
            }
        });
        long complete = 0, right_place = 0;
        for (long w = 0; w < K; w++) {
            s.sync.begin_wait(w);
            s.holding.store(-1);
            bool ok = true;
            for (int q = 0; q < R; q++) ok = ok && s.sync.wait_submitted(q, w);     // (the simulation polls every rank: a shard is there once its rank has submitted)
            if (!ok) break;
            s.holding.store(w);
            const std::vector<long>& set = s.buf[(size_t)s.plan.collector_index(w)][(size_t)Plan::slot(w)];
            bool all = true;
            for (int q = 0; q < R; q++) all = all && set[(size_t)q] == w;
            complete += all ? 1 : 0;
            right_place += s.plan.collector_rank(w) == (rot ? (int)(w % R) : 0) ? 1 : 0;
            if (w % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));      // a slow consumer: the ranks run into the back-pressure
        }
        for (auto& t : th) t.join();
        CHECK(complete == K); CHECK(right_place == K); CHECK(s.overwrites.load() == 0);
    }
    return fails;
}

static int test_abort() {
    const int R = 4;
    std::vector<int> dev = {0, 1, 2, 3};
    Sim s(R, 1000, true, dev);
    std::atomic<int> returned{0};
    std::vector<std::thread> th;
    for (int r = 0; r < R; r++) th.emplace_back([&s, r, &returned] {
        for (long k = 0; k < s.K; k++) {
            if (r == 2 && k == 3) break;                                // rank 2 fails at block 3 and never submits again
            if (!s.sync.begin_submit(k)) break;
            s.sync.end_submit(r, k);
        }
        returned++;
    });
    std::thread coll([&s, &returned] {
        for (long w = 0; w < s.K; w++) {
            s.sync.begin_wait(w);
            bool ok = true;
            for (int q = 0; q < s.R && ok; q++) ok = s.sync.wait_submitted(q, w);
            if (!ok) break;
        }
        returned++;
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(returned.load() == 1);                                        // only the failed rank is out: everybody else waits (collector for block 3, ranks for buffers)
    CHECK(!s.sync.abort());
    CHECK(s.sync.abort());                                              // a second abort is a no-op
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& t : th) t.join();
    coll.join();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    CHECK(returned.load() == R + 1); CHECK(ms < 2000.0);
    CHECK(!s.sync.begin_submit(0));                                     // and from here on
    return fails;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "plan";
    if (mode == "plan") test_plan();
    else if (mode == "run") test_run(argc > 2 ? atoi(argv[2]) : 4, argc > 3 ? atol(argv[3]) : 200);
    else if (mode == "abort") test_abort();
    else return 2;
    printf("{\"mode\": \"%s\", \"failed_checks\": %d}\n", mode.c_str(), fails);
    return fails ? 1 : 0;
}
