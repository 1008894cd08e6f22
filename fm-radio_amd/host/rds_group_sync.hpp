// RDS block / group synchroniser on the host: consumes the Manchester-decoded byte stream the GPU
// produces (fmd_get_rds_bytes, 16 bytes at a time like the reference's DifferentialManchesterDecoder buffer) and emits
// 4-block groups.  Behaviour follows the reference's RDS_Group_Sync state machine (reference
// src/rds_decoder/rds_group_sync.cpp:29-127: FINDING_SYNC on offset word A with a zero CRC-10 syndrome, READ_BLOCK with
// per-block offset words A,B,C|C',D, single-bit error correction from a syndrome table crc10.cpp:28-60, resync after 3
// consecutive groups with errors) and its coding constants (rds_constants.h:6-40).  1187.5 bit/s per station: host work.
#pragma once

#include <cstdint>
#include <cstddef>
#include <functional>
#include <unordered_map>

namespace fmd_host {

struct RDS_Block { uint16_t data; int block_type; bool is_valid; };   // block_type: 0 A, 1 B, 2 C, 3 C', 4 D
struct RDS_Group { RDS_Block blocks[4]; };

class RDS_Group_Sync_Host {
    static constexpr int kBlockBits = 26, kCrcBits = 10;
    static constexpr uint16_t kPoly = 0x1B9;  // x^10 + x^8 + x^7 + x^5 + x^4 + x^3 + 1 (x^10 implicit)
    static constexpr uint16_t kOffsets[5] = {0x0FC, 0x198, 0x168, 0x350, 0x1B4};  // A, B, C, C', D
    uint32_t block_buf = 0;
    int block_bits = 0;
    RDS_Group group{};
    int curr_block = 0, block_errors = 0;
    int groups_desync = 0, bits_desync = 0;
    bool finding_sync = true;
    std::unordered_map<uint16_t, uint32_t> error_patterns;
    std::function<void(const RDS_Group&)> on_group;
    std::function<void(int)> on_lock;

public:
    static uint16_t crc10(uint32_t x) {
        uint16_t reg = 0;
        for (int i = 0; i < kBlockBits; i++) {
            const uint16_t bit = (uint16_t)((x >> (kBlockBits - 1)) & 1u);
            x <<= 1;
            reg = (uint16_t)((reg << 1) | bit);
            if (reg & (1u << kCrcBits)) reg ^= kPoly | (1u << kCrcBits);
        }
        return reg & 0x3FF;
    }
    RDS_Group_Sync_Host() {
        for (int i = kCrcBits; i < kBlockBits; i++) error_patterns[crc10(1u << i)] = 1u << i;
        for (int i = 0; i < kCrcBits; i++) error_patterns[crc10(1u << i)] = 1u << i;
    }
    void OnGroup(std::function<void(const RDS_Group&)> fn) { on_group = std::move(fn); }
    void OnLock(std::function<void(int bits_searched)> fn) { on_lock = std::move(fn); }
    bool IsLocked() const { return !finding_sync; }

    void Process(const uint8_t* x, size_t n_bytes) {
        const size_t n_bits = n_bytes * 8;
        for (size_t i = 0; i < n_bits; i++) {
            const uint32_t bit = (x[i / 8] >> (7 - (i % 8))) & 1u;
            block_buf = ((block_buf << 1) | bit) & ((1u << kBlockBits) - 1u);
            if (finding_sync) {
                bits_desync++;
                if (crc10(block_buf ^ kOffsets[0]) != 0) { bits_desync++; continue; }
                if (on_lock) on_lock(bits_desync);
                finding_sync = false;
                bits_desync = 0;
                block_bits = 0;
                push_block(block_buf);
                continue;
            }
            if (++block_bits != kBlockBits) continue;
            block_bits = 0;
            push_block(block_buf);
            if (curr_block < 4) continue;
            if (on_group) on_group(group);
            const int errors = block_errors;
            curr_block = 0;
            block_errors = 0;
            if (errors == 0) { groups_desync = 0; continue; }
            if (++groups_desync >= 3) { finding_sync = true; groups_desync = 0; }
        }
    }

private:
    bool attempt(uint32_t x, int type, RDS_Block& b) {
        x ^= kOffsets[type];
        uint32_t corrected = x;
        bool valid = false;
        const uint16_t syndrome = crc10(x);
        if (syndrome == 0) valid = true;
        else {
            auto it = error_patterns.find(syndrome);
            if (it != error_patterns.end() && crc10(x ^ it->second) == 0) { corrected = x ^ it->second; valid = true; }
        }
        b.block_type = type;
        b.data = (uint16_t)((corrected >> kCrcBits) & 0xFFFF);
        b.is_valid = valid;
        return valid;
    }
    void push_block(uint32_t x) {
        RDS_Block& b = group.blocks[curr_block];
        b.is_valid = false;
        switch (curr_block) {
            case 0: attempt(x, 0, b); break;
            case 1: attempt(x, 1, b); break;
            case 2: if (!attempt(x, 2, b)) attempt(x, 3, b); break;
            case 3: attempt(x, 4, b); break;
            default: break;
        }
        curr_block++;
        if (!b.is_valid) block_errors++;
    }
};

}  // namespace fmd_host
