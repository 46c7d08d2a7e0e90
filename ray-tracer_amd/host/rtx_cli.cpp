// rtx_cli.cpp — `rtx <scene.yaml>`: the reference's CLI (main.rs:11-28) on the MI355X path.
// `rtx <scene.yaml> --all-cameras` additionally renders the cameras the reference parses but ignores (N4).
#include <cstdio>
#include <cstring>

#include "../../include/rtx_host.h"

int main(int argc, char** argv) {
    if (argc < 2) {
        std::printf("Need to provide filename of scene description as argument\n");   // main.rs:16
        return 0;
    }
    const bool all = argc > 2 && std::strcmp(argv[2], "--all-cameras") == 0;
    int rc = all ? rtxh_render_file_all(argv[1], nullptr, 0, nullptr) : rtxh_render_file(argv[1], nullptr, 0);
    if (rc != RTX_OK) {
        std::fprintf(stderr, "rtx: %s (%s)\n", rtx_strerror(rc), rtxh_last_error());
        return 1;
    }
    return 0;
}
