// Test program (CPU): the file rendezvous of capital_amd/bench/launch.h run by real processes against the shim's
// capi_comm_unique_id (random per call).  Prints the 128-byte id this rank ended up with; the test compares ranks.
#include "../../capital_amd/bench/launch.h"

int main() {
  try {
    const int rank = capital_bench::env_int("RANK", 0), size = capital_bench::env_int("WORLD_SIZE", 1);
    const char* path = getenv("CAPITAL_UID_FILE");
    unsigned char uid[128] = {0};
    capital_bench::rendezvous_uid(path, rank, size, uid, (double)capital_bench::env_int("CAPITAL_RENDEZVOUS_TIMEOUT_S", 20));
    if (rank == 0 && !getenv("CAPITAL_KEEP_UID_FILES")) capital_bench::rendezvous_cleanup(path, size);
    for (int i = 0; i < 128; ++i) printf("%02x", uid[i]);
    printf("\n");
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "rendezvous_main: %s\n", e.what());
    return 3;
  }
}
