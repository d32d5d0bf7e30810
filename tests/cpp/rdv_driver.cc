// tests/cpp/rdv_driver.cc -- forks `world` ranks that meet through the data-parallel rendezvous (csrc/bp_rdv.h: shared-memory
// block, host barrier, all-gather) without a GPU; built with -fsanitize=address,undefined by tests/test_sanitizers.py.
//   rdv_driver <key> <world> [timeout_s]      exit code 0: every rank saw every rank's record and left cleanly
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <vector>
#include "../../dnn-for-speech-enhancement_amd/csrc/bp_rdv.h"

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int world = atoi(argv[2]);
    const double tmo = argc > 3 ? atof(argv[3]) : 20.0;
    int rank = 0;
    std::vector<pid_t> kids;
    for (int k = 1; k < world; ++k) { const pid_t c = fork(); if (c == 0) { rank = k; kids.clear(); break; } kids.push_back(c); }
    bp_rdv *r = nullptr;
    if (rdv_open(argv[1], world, rank, tmo, &r) != 0) { printf("rank %d: %s\n", rank, g_rdv_err.c_str()); return 3; }
    int bad = 0;
    for (int round = 0; round < 5; ++round) {
        double mine = 100.0 * round + rank, all[BP_RDV_MAXRANKS];
        if (rdv_allgather(r, &mine, sizeof(mine), all) != 0) { printf("rank %d: %s\n", rank, g_rdv_err.c_str()); bad = 1; break; }
        for (int p = 0; p < world; ++p) if (all[p] != 100.0 * round + p) bad = 1;
        if (rdv_barrier(r) != 0) { bad = 1; break; }
    }
    rdv_close(r, bad != 0);
    if (rank != 0) _exit(bad ? 4 : 0);
    for (pid_t c : kids) { int st = 0; waitpid(c, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
    return bad ? 5 : 0;
}
