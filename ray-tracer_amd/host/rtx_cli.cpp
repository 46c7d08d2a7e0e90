// rtx_cli.cpp — `rtx <scene.yaml>`: the reference's CLI (main.rs:11-28) on the MI355X path.
// `rtx <scene.yaml> --all-cameras` additionally renders the cameras the reference parses but ignores (N4).
// `rtx <scene.yaml> --gpus N` renders cameras[0] on N GPUs (0 = all): tile bands per GPU + RCCL all-gather.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rtx_host.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::printf("Need to provide filename of scene description as argument\n");   // main.rs:16
        return 0;
    }
    bool all = false; int gpus = -1;
    for (int i = 2; i < argc; i++) {
        if (std::strcmp(argv[i], "--all-cameras") == 0) all = true;
        else if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else { std::fprintf(stderr, "rtx: unknown argument %s\n", argv[i]); return 2; }
    }
    int rc = all ? rtxh_render_file_all(argv[1], nullptr, 0, nullptr)
           : gpus >= 0 ? rtxh_render_file_gpus(argv[1], nullptr, gpus) : rtxh_render_file(argv[1], nullptr, 0);
    if (rc != RTX_OK) {
        std::fprintf(stderr, "rtx: %s (%s)\n", rtx_strerror(rc), rtxh_last_error());
        return 1;
    }
    return 0;
}
