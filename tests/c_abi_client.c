/* tests/c_abi_client.c -- a plain C99 client of include/zkhip.h: proves the header is C (not C++), that the
 * library links from C, and runs the host-side verifier on a (vk, proof) pair given as files.
 * usage: c_abi_client <vk.json> <proof.json>   -> prints "accepted" / "rejected", exit 0 / 1 (2 on error) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zkhip.h"

static char *slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    char *b = (char *)malloc((size_t)n + 1);
    if (b && fread(b, 1, (size_t)n, f) != (size_t)n) { free(b); b = NULL; }
    if (b) b[n] = 0;
    fclose(f);
    return b;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <vk.json> <proof.json>\n", argv[0]); return 2; }
    char *vk = slurp(argv[1]), *proof = slurp(argv[2]);
    if (!vk || !proof) { fprintf(stderr, "cannot read inputs\n"); return 2; }
    printf("%s; zk_domain_size(1048574, 1) = %u\n", zk_version(), zk_domain_size(1048574u, 1u));
    int ok = 0;
    int rc = zk_verify(vk, proof, &ok);
    if (rc != ZK_OK) { fprintf(stderr, "zk_verify: %s (%s)\n", zk_strerror(rc), zk_last_error()); return 2; }
    bool same = ethsnarks_verify(vk, proof);
    if ((ok == 1) != same) return 2;
    puts(ok ? "accepted" : "rejected");
    free(vk); free(proof);
    return ok ? 0 : 1;
}
