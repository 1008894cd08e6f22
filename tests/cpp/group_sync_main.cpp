// rds bytes file -> group log in the reference's stderr format ("[group] [AAAA BBBB CCCC DDDD]", "----" for invalid blocks)
#include <cstdio>
#include <vector>
#include "rds_group_sync.hpp"
int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    std::vector<uint8_t> data; uint8_t buf[16]; size_t n;
    fmd_host::RDS_Group_Sync_Host sync;
    sync.OnLock([](int bits) { printf("Locked onto block A after %d bits\n", bits); });
    sync.OnGroup([](const fmd_host::RDS_Group& g) {
        printf("[group] [");
        for (int i = 0; i < 4; i++) {
            if (g.blocks[i].is_valid) printf("%04X", g.blocks[i].data); else printf("----");
            printf(i < 3 ? " " : "]\n");
        }
    });
    while ((n = fread(buf, 1, 16, f)) == 16) sync.Process(buf, 16);   // 16 bytes at a time, like the reference's chain
    fclose(f);
    return 0;
}
