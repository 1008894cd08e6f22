// Test driver of fm-radio_amd/host/multi_gpu_host.hpp + libfmdgather.so (include/fmdemod_gather.h).
//
//   multi_gpu_main <captures.u8> <n_ranks> <stations_per_rank> <block_size> <fs> <n_blocks> <f32|pcm16> <loopback 0|1> [fast] [rotate]
//       rotate: FMD_GATHER_ROTATE, block k is collected on rank k mod n_ranks (needs one GPU per rank)
//       captures.u8 holds [n_ranks * stations_per_rank][n_blocks * block_size][2] u8; rank r gets device r when the box has that
//       many GPUs, otherwise every rank shares device 0 (copy hand-over; with loopback = 1 and one rank the shard travels through
//       ncclSend / ncclRecv to the self peer).
//   Pass 1, lock-step: after every block the gathered audio / RDS bytes / counts on the collector must equal what every rank's own
//   fmd_get_audio / fmd_get_rds_bytes return (PCM16: the reference scraper's conversion of it, fm_scraper.cpp:79-82).
//   Pass 2, fresh handles, two blocks in flight ahead of the collector: the gathered blocks must equal pass 1's bit for bit.
//   failrank: a third pass in which one rank's submission fails: Collect() throws and the host tears down (abort path).
//   misuse: the C ABI directly, one rank: a second fmd_gather_submit without a new block and a skipped block are REFUSED
//       (FMD_ERR_ARG; each on a fresh gather); a handle reset between two blocks (fmd_outputs_epoch changes) is re-based and gathers on.
//       (A handle under fmd_set_output_lag whose stages are put off — 1024 stations and more — shows block k - 1: the same refusal, covered
//       without a GPU by tests/cpp/gather_plan_main.cpp; batches this small queue every block at submission and gather normally.)
//   Prints one JSON line; exit code 0 only if everything matched.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "multi_gpu_host.hpp"

using fmd_host::MultiGpuHost;

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 5; } } while (0)

struct Block { std::vector<char> audio; std::vector<uint8_t> bytes; std::vector<int> counts; };

int main(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "usage: multi_gpu_main <captures.u8> <n_ranks> <stations_per_rank> <block> <fs> <n_blocks> <f32|pcm16> <loopback> [fast]\n"); return 1; }
    const std::string path = argv[1];
    const int R = atoi(argv[2]), C = atoi(argv[3]), N = atoi(argv[4]), fs = atoi(argv[5]), nb = atoi(argv[6]);
    const bool pcm = std::string(argv[7]) == "pcm16", loopback = atoi(argv[8]) != 0;
    bool fast = false, rotate = false, failrank = false, misuse = false;
    for (int i = 9; i < argc; i++) { fast = fast || std::string(argv[i]) == "fast"; rotate = rotate || std::string(argv[i]) == "rotate"; failrank = failrank || std::string(argv[i]) == "failrank"; misuse = misuse || std::string(argv[i]) == "misuse"; }
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return 2;
    std::vector<uint8_t> data((size_t)R * C * nb * N * 2);
    if (fread(data.data(), 1, data.size(), fp) != data.size()) return 2;
    fclose(fp);
    int ndev = 0;
    HIPC(hipGetDeviceCount(&ndev));
    std::vector<int> devs((size_t)R);
    for (int r = 0; r < R; r++) devs[(size_t)r] = ndev >= R ? r : 0;
    try {
        // every rank's blocks on its device: [nb][C][N][2]
        std::vector<std::vector<uint8_t*>> d_in((size_t)R, std::vector<uint8_t*>((size_t)nb, nullptr));
        const size_t blk = (size_t)C * N * 2;
        std::vector<uint8_t> tmp(blk);
        for (int r = 0; r < R; r++) {
            HIPC(hipSetDevice(devs[(size_t)r]));
            for (int b = 0; b < nb; b++) {
                for (int c = 0; c < C; c++)
                    std::memcpy(tmp.data() + (size_t)c * N * 2, data.data() + (((size_t)(r * C + c) * nb + b) * N) * 2, (size_t)N * 2);
                HIPC(hipMalloc(reinterpret_cast<void**>(&d_in[(size_t)r][(size_t)b]), blk));
                HIPC(hipMemcpy(d_in[(size_t)r][(size_t)b], tmp.data(), blk, hipMemcpyHostToDevice));
            }
        }
        const unsigned dflags = fast ? FMD_FLAG_FAST_MATH : 0u, gflags = (loopback ? FMD_GATHER_LOOPBACK_RCCL : 0u) | (rotate ? FMD_GATHER_ROTATE : 0u);
        const int fmt = pcm ? FMD_GATHER_PCM16 : FMD_GATHER_F32;
        long mismatches = 0;
        std::vector<Block> pass1((size_t)nb);
        size_t remote = 0;
        int n_audio = 0, cap = 0;
        auto fetch = [&](const MultiGpuHost::Views& v, Block& out, int n_audio_) -> int {
            const int root_dev = v.device;
            const size_t ab = (size_t)R * C * n_audio_ * 2 * (pcm ? 2 : 4);
            out.audio.resize(ab); out.bytes.resize((size_t)R * C * v.rds_cap); out.counts.resize((size_t)R * C);
            HIPC(hipSetDevice(root_dev));
            HIPC(hipMemcpy(out.audio.data(), v.audio, ab, hipMemcpyDeviceToHost));
            HIPC(hipMemcpy(out.bytes.data(), v.rds_bytes, out.bytes.size(), hipMemcpyDeviceToHost));
            HIPC(hipMemcpy(out.counts.data(), v.rds_counts, out.counts.size() * sizeof(int), hipMemcpyDeviceToHost));
            return 0;
        };
        {
            MultiGpuHost host(devs, C, N, fs, dflags, fmt, gflags, 0);
            n_audio = host.Rates().n_audio;
            remote = host.RemoteBytesPerBlock();
            std::vector<float> a((size_t)C * n_audio * 2);
            for (int b = 0; b < nb; b++) {
                std::vector<const uint8_t*> ptrs((size_t)R);
                for (int r = 0; r < R; r++) ptrs[(size_t)r] = d_in[(size_t)r][(size_t)b];
                host.SubmitU8(ptrs);
                const MultiGpuHost::Views v = host.Collect();
                cap = v.rds_cap;
                if (int rc = fetch(v, pass1[(size_t)b], n_audio)) return rc;
                for (int r = 0; r < R; r++) {      // what the rank's own handle holds for this block
                    if (fmd_get_audio(host.Handle(r), a.data()) != FMD_OK) return 6;
                    std::vector<uint8_t> by((size_t)C * cap); std::vector<int> cn((size_t)C);
                    if (fmd_get_rds_bytes(host.Handle(r), by.data(), cap, cn.data()) != FMD_OK) return 6;
                    for (size_t i = 0; i < a.size(); i++) {
                        if (pcm) {
                            const int16_t want = (int16_t)(int)(a[i] * (32767.0f * 0.95f));
                            mismatches += reinterpret_cast<const int16_t*>(pass1[(size_t)b].audio.data())[(size_t)r * a.size() + i] != want;
                        } else mismatches += std::memcmp(&reinterpret_cast<const float*>(pass1[(size_t)b].audio.data())[(size_t)r * a.size() + i], &a[i], 4) != 0;
                    }
                    for (int c = 0; c < C; c++) {
                        mismatches += pass1[(size_t)b].counts[(size_t)r * C + c] != cn[(size_t)c];
                        mismatches += std::memcmp(pass1[(size_t)b].bytes.data() + ((size_t)r * C + c) * cap, by.data() + (size_t)c * cap, (size_t)cn[(size_t)c]) != 0;
                    }
                }
            }
        }
        long bytes_total = 0;
        for (const Block& b : pass1) for (int c : b.counts) bytes_total += c;
        long mismatches2 = 0;
        {
            MultiGpuHost host(devs, C, N, fs, dflags, fmt, gflags, 0);
            int submitted = 0, collected = 0;
            Block got;
            while (collected < nb) {
                while (submitted < nb && submitted - collected < 3) {     // two blocks in flight beyond the one being collected
                    std::vector<const uint8_t*> ptrs((size_t)R);
                    for (int r = 0; r < R; r++) ptrs[(size_t)r] = d_in[(size_t)r][(size_t)submitted];
                    host.SubmitU8(ptrs);
                    submitted++;
                }
                const MultiGpuHost::Views v = host.Collect();
                if (int rc = fetch(v, got, n_audio)) return rc;
                const Block& w = pass1[(size_t)collected];
                mismatches2 += got.audio != w.audio;
                mismatches2 += got.counts != w.counts;
                for (size_t s = 0; s < w.counts.size(); s++) mismatches2 += std::memcmp(got.bytes.data() + s * cap, w.bytes.data() + s * cap, (size_t)w.counts[s]) != 0;
                collected++;
            }
        }
        // Pass 3 (failrank): the last rank's submission of block 2 fails (a null block): Collect() must throw that rank's error instead of
        // waiting for a block that never comes, and the host's destructor (abort, join, fmd_gather_destroy: communicators aborted before the
        // rank streams are drained) must return.
        double teardown_ms = -1.0; int threw = 0;
        if (failrank) {
            const auto t0 = std::chrono::steady_clock::now();
            {
                MultiGpuHost host(devs, C, N, fs, dflags, fmt, gflags, 0);
                try {
                    for (int b = 0; b < nb; b++) {
                        std::vector<const uint8_t*> ptrs((size_t)R);
                        for (int r = 0; r < R; r++) ptrs[(size_t)r] = (b == 2 && r == R - 1) ? nullptr : d_in[(size_t)r][(size_t)b];
                        host.SubmitU8(ptrs);
                        (void)host.Collect();
                    }
                } catch (const std::exception& e) { threw = std::string(e.what()).find("fmd_submit") != std::string::npos ? 1 : -1; }
            }
            teardown_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (threw != 1) mismatches2 += 1000;
        }
        // Pass 4 (misuse): what fmd_gather_submit must refuse, and the one renumbering it must follow
        int misuse_ok = -1;
        if (misuse) {
            misuse_ok = 1;
            auto scenario = [&](int kind) -> bool {      // 0: repeated submit, 1: skipped block, 3: reset between blocks (must WORK)
                HIPC(hipSetDevice(devs[0]));
                fmd_handle h = nullptr;
                fmd_config cfg{C, N, fs, devs[0], dflags};
                if (fmd_create(&cfg, &h) != FMD_OK) return false;
                const int one_dev[1] = {devs[0]};
                fmd_gather_config gc{1, one_dev, 0, fmt, gflags & FMD_GATHER_LOOPBACK_RCCL};
                fmd_gather g = nullptr;
                if (fmd_gather_create(&gc, &h, &g) != FMD_OK) { fmd_destroy(h); return false; }
                const void* va; const uint8_t* vb; const int* vc; int vcap;
                bool ok = true;
                auto block = [&](int b) { return fmd_submit_u8_dev(h, d_in[0][(size_t)b], C, N, nullptr) == FMD_OK; };
                if (kind == 0) {
                    ok = ok && block(0) && fmd_gather_submit(g, 0) == FMD_OK && fmd_gather_wait(g, &va, &vb, &vc, &vcap) == FMD_OK;
                    ok = ok && fmd_gather_submit(g, 0) == FMD_ERR_ARG;                       // the same block again
                } else if (kind == 1) {
                    ok = ok && block(0) && block(1) && fmd_gather_submit(g, 0) == FMD_ERR_ARG;      // block 1 is not the gather's block 0
                } else {
                    ok = ok && block(0) && fmd_gather_submit(g, 0) == FMD_OK && fmd_gather_wait(g, &va, &vb, &vc, &vcap) == FMD_OK;
                    ok = ok && fmd_reset(h) == FMD_OK;
                    ok = ok && block(1) && fmd_gather_submit(g, 0) == FMD_OK && fmd_gather_wait(g, &va, &vb, &vc, &vcap) == FMD_OK;     // the handle's block 0 again, the gather's block 1
                    ok = ok && block(2) && fmd_gather_submit(g, 0) == FMD_OK && fmd_gather_wait(g, &va, &vb, &vc, &vcap) == FMD_OK;
                }
                if (kind != 3) fmd_gather_abort(g);
                fmd_gather_destroy(g);
                fmd_destroy(h);
                return ok;
            };
            for (int kind : {0, 1, 3}) if (!scenario(kind)) { misuse_ok = 0; fprintf(stderr, "misuse scenario %d\n", kind); }
            if (misuse_ok != 1) mismatches2 += 1000;
        }
        printf("{\"misuse_handled\": %d, \"failed_rank_threw\": %d, \"failed_rank_run_ms\": %.1f, ", misuse_ok, threw, teardown_ms);
        printf("\"ranks\": %d, \"devices\": %d, \"stations_per_rank\": %d, \"blocks\": %d, \"format\": \"%s\", \"loopback_rccl\": %s, \"rotate\": %s, \"lockstep_mismatches\": %ld, "
               "\"pipelined_mismatches\": %ld, \"rds_bytes_gathered\": %ld, \"remote_bytes_per_block\": %zu}\n",
               R, ndev, C, nb, pcm ? "pcm16" : "f32", loopback ? "true" : "false", rotate ? "true" : "false", mismatches, mismatches2, bytes_total, remote);
        return (mismatches == 0 && mismatches2 == 0 && bytes_total > 0) ? 0 : 3;
    } catch (const std::exception& e) {
        fprintf(stderr, "multi_gpu_main: %s\n", e.what());
        return 4;
    }
}
