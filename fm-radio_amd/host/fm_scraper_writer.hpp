// Scraper-compatible output files (host side, no GPU involved): the 16-bit stereo WAV and raw RDS byte files the
// reference's headless `fm_demod_scraper` writes (reference src/fm_scraper.cpp:55-200), so audio can be auditioned and
// the RDS bytes fed to the reference's `rds_decode` tool.
//
// Kept byte-compatible with the reference, quirks included:
//  * samples are float * (32767 * 0.95f), truncated toward zero to int16 (fm_scraper.cpp:79-82, Frame<T>::operator Frame<U>)
//  * the RIFF / data chunk sizes are rewritten after every block, and count FRAMES written, not bytes
//    (`total_bytes_written += int(nb_written)` with nb_written = frames, fm_scraper.cpp:84-89,152-166)
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace fmd_host {

class Audio_WAV_Writer {
    FILE* fp = nullptr;
    int fs = 0;
    int total_written = 0;
    std::vector<int16_t> convert;
#pragma pack(push, 1)
    struct WavHeader {
        char ChunkID[4]; int32_t ChunkSize; char Format[4];
        char Subchunk1ID[4]; int32_t Subchunk1Size; int16_t AudioFormat; int16_t NumChannels; int32_t SampleRate; int32_t ByteRate;
        int16_t BlockAlign; int16_t BitsPerSample;
        char Subchunk2ID[4]; int32_t Subchunk2Size;
    };
#pragma pack(pop)
    void update_header() {
        const int32_t sub2 = total_written, chunk = 36 + sub2;
        fseek(fp, 4, SEEK_SET); fwrite(&chunk, 4, 1, fp);
        fseek(fp, 40, SEEK_SET); fwrite(&sub2, 4, 1, fp);
        fseek(fp, 0, SEEK_END);
    }
public:
    explicit Audio_WAV_Writer(const std::string& path, int sample_rate = 32000) : fs(sample_rate) {
        fp = fopen(path.c_str(), "wb+");
        if (!fp) return;
        WavHeader h;
        memcpy(h.ChunkID, "RIFF", 4); memcpy(h.Format, "WAVE", 4); memcpy(h.Subchunk1ID, "fmt ", 4); memcpy(h.Subchunk2ID, "data", 4);
        h.Subchunk1Size = 16; h.AudioFormat = 1; h.NumChannels = 2; h.SampleRate = fs; h.BitsPerSample = 16;
        h.ByteRate = h.SampleRate * h.NumChannels * h.BitsPerSample / 8;
        h.BlockAlign = (int16_t)(h.NumChannels * h.BitsPerSample / 8);
        h.Subchunk2Size = 0; h.ChunkSize = 36;
        fwrite(&h, sizeof(h), 1, fp);
    }
    ~Audio_WAV_Writer() { if (fp) { update_header(); fclose(fp); } }
    bool ok() const { return fp != nullptr; }
    // frames: interleaved L,R floats (the demodulator's audio block)
    void on_audio_data(const float* frames, size_t n_frames) {
        if (!fp) return;
        const float scale = 32767.0f * 0.95f;
        convert.resize(2 * n_frames);
        for (size_t i = 0; i < 2 * n_frames; i++) convert[i] = (int16_t)(int32_t)(frames[i] * scale);
        const size_t nb = fwrite(convert.data(), 4, n_frames, fp);
        total_written += (int)nb;
        update_header();
    }
};

class RDS_Bytes_Writer {
    FILE* fp = nullptr;
public:
    explicit RDS_Bytes_Writer(const std::string& path) { fp = fopen(path.c_str(), "wb+"); }
    ~RDS_Bytes_Writer() { if (fp) fclose(fp); }
    bool ok() const { return fp != nullptr; }
    void on_rds_bytes(const uint8_t* data, size_t n) { if (fp) fwrite(data, 1, n, fp); }
};

}  // namespace fmd_host
