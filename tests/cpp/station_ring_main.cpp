// Test / bench driver of fm-radio_amd/host/station_ring.hpp.
//
//   station_ring_main check <captures.u8> <outdir> <n_stations> <block_size> <fs> <n_blocks>
//       captures.u8 holds [n_stations][n_blocks * block_size (+ a partial tail)][2] u8.  Every station's capture is pushed in
//       ragged pieces of a station-specific size, stations interleaved, the way C independent receivers would deliver them;
//       per-station audio and RDS bytes are written to <outdir>/audio_<c>.f32 / rds_<c>.u8 for the Python side to compare
//       with the oracle.
//   station_ring_main bench <n_stations> <block_size> <fs> <n_blocks> <threads> [fast]
//       host-fed throughput: `threads` producer threads (stations t, t + threads, ...) push block-sized pieces of synthetic u8
//       IQ as fast as the ring takes them; prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "station_ring.hpp"

using fmd_host::StationRing;

static int run_check(int argc, char** argv) {
    if (argc < 8) return 1;
    const std::string path = argv[2], out = argv[3];
    const int C = atoi(argv[4]), N = atoi(argv[5]), fs = atoi(argv[6]), nb = atoi(argv[7]);
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return 2;
    fseek(fp, 0, SEEK_END); const size_t bytes = (size_t)ftell(fp); fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> data(bytes);
    if (fread(data.data(), 1, bytes, fp) != bytes) return 2;
    fclose(fp);
    const size_t per = bytes / 2 / (size_t)C;                 // samples per station in the file (whole blocks + tail)
    StationRing ring(C, N, fs);
    std::vector<FILE*> fa((size_t)C), fr((size_t)C);
    for (int c = 0; c < C; c++) {
        fa[(size_t)c] = fopen((out + "/audio_" + std::to_string(c) + ".f32").c_str(), "wb");
        fr[(size_t)c] = fopen((out + "/rds_" + std::to_string(c) + ".u8").c_str(), "wb");
    }
    ring.OnAudioBlock([&](int c, const fmd_host::Frame* x, size_t n, int) { fwrite(x, sizeof(fmd_host::Frame), n, fa[(size_t)c]); });
    ring.On_RDS_Bytes([&](int c, const uint8_t* x, size_t n) { fwrite(x, 1, n, fr[(size_t)c]); });
    std::vector<size_t> pos((size_t)C, 0);
    bool any = true;
    long stalls = 0;
    while (any) {
        any = false;
        for (int c = 0; c < C; c++) {
            const size_t piece = 1000 + 37 * (size_t)(c % 97) + 4096 * (size_t)(c % 3);   // ragged, station-specific, unrelated to the block size
            const size_t left = per - pos[(size_t)c];
            if (!left) continue;
            const size_t n = left < piece ? left : piece;
            const size_t took = ring.Push(c, data.data() + ((size_t)c * per + pos[(size_t)c]) * 2, n);
            pos[(size_t)c] += took;
            if (took < n) stalls++;
            any = true;
        }
        ring.Poll();
    }
    ring.Flush();
    for (int c = 0; c < C; c++) { fclose(fa[(size_t)c]); fclose(fr[(size_t)c]); }
    printf("{\"blocks_delivered\": %ld, \"expected\": %d, \"push_stalls\": %ld}\n", ring.BlocksDelivered(), nb, stalls);
    return ring.BlocksDelivered() == nb ? 0 : 3;
}

static int run_bench(int argc, char** argv) {
    if (argc < 7) return 1;
    const int C = atoi(argv[2]), N = atoi(argv[3]), fs = atoi(argv[4]), nb = atoi(argv[5]), T = atoi(argv[6]);
    const bool fast = argc > 7 && std::string(argv[7]) == "fast";
    StationRing ring(C, N, fs, fast ? FMD_FLAG_FAST_MATH : 0u, 4);
    size_t audio_frames = 0;
    ring.OnAudioBlock([&](int, const fmd_host::Frame*, size_t n, int) { audio_frames += n; });
    // one block of plausible u8 IQ per thread (values do not matter for the timing; kept away from the all-127 dead input)
    std::vector<std::vector<uint8_t>> src((size_t)T, std::vector<uint8_t>((size_t)N * 2));
    for (int t = 0; t < T; t++) for (size_t i = 0; i < src[(size_t)t].size(); i++) src[(size_t)t][i] = (uint8_t)(27 + ((i * 2654435761u + (unsigned)t * 97u) >> 13) % 200);
    std::atomic<bool> go{false};
    auto producer = [&](int t, int blocks) {
        while (!go.load()) std::this_thread::yield();
        for (int b = 0; b < blocks; b++)
            for (int c = t; c < C; c += T) {
                size_t done = 0;
                while (done < (size_t)N) {
                    const size_t took = ring.Push(c, src[(size_t)t].data() + 2 * done, (size_t)N - done);
                    done += took;
                    if (!took) std::this_thread::yield();          // the ring is `depth` blocks ahead of the GPU: the owner's Poll() frees it
                }
            }
    };
    auto run = [&](int blocks) {
        std::vector<std::thread> th;
        go.store(false);
        const long target = ring.BlocksDelivered() + blocks;
        for (int t = 0; t < T; t++) th.emplace_back(producer, t, blocks);
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true);
        while (ring.BlocksDelivered() < target) ring.Poll();
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (auto& x : th) x.join();
        return el;
    };
    run(4);                                                         // warm-up: pinned pages touched, loops started
    const double el = run(nb);
    const double msa = (double)C * N * nb / el / 1e6;
    printf("{\"host_fed_msa_per_s\": %.1f, \"stations\": %d, \"block_size\": %d, \"fs\": %d, \"blocks\": %d, \"producer_threads\": %d, \"seconds\": %.4f, "
           "\"h2d_gb_per_s\": %.2f, \"d2h_gb_per_s\": %.2f, \"mode\": \"%s\", \"audio_frames_delivered\": %zu}\n",
           msa, C, N, fs, nb, T, el, msa * 2e6 / 1e9, (double)C * ring.Rates().n_audio * 8.0 * nb / el / 1e9, fast ? "fast_math" : "exact", audio_frames);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: station_ring_main check|bench ...\n"); return 1; }
    try {
        return std::string(argv[1]) == "bench" ? run_bench(argc, argv) : run_check(argc, argv);
    } catch (const std::exception& e) {
        fprintf(stderr, "station_ring_main: %s\n", e.what());
        return 4;
    }
}
