// ref_cmd_driver.cpp -- TEST INFRASTRUCTURE.  Command-line driver around the reference's own sub-commands that need no
// sequence input (src/subCommand.cpp, compiled unmodified from where it lies; it includes no RabbitFX header):
//   command_info   (src/subCommand.cpp:70-147)      command_union (:307-543)
//   command_sub    (:545-794)                       command_merge (:796-892)
// linked with oracle/_ref/ref_sketch.o (the RabbitFX-free lines of sketch.cpp, see oracle/Makefile) and the unmodified
// dist.cpp / common.cpp / shuffle.cpp.  This file contains no reference code; it only calls those four functions, so
// that `rabbit_kssd info|union|sub|merge` can be pinned against what the reference itself writes
// (tests/golden/make_f4_golden.py -> tests/golden/f4).
#include <cstdio>
#include <cstdlib>
#include <string>

#include "subCommand.h"

int main(int argc, char **argv)
{
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "info" && argc == 5) { command_info(argv[2], atoi(argv[3]) != 0, argv[4]); return 0; }
    if (cmd == "union" && argc == 5) { command_union(argv[2], argv[3], atoi(argv[4])); return 0; }
    if (cmd == "sub" && argc == 6) { command_sub(argv[2], argv[3], argv[4], atoi(argv[5])); return 0; }
    if (cmd == "merge" && argc == 5) { command_merge(argv[2], argv[3], atoi(argv[4])); return 0; }
    fprintf(stderr, "usage:\n  ref_cmd_driver info in.sketch detail out.txt\n  ref_cmd_driver union in.sketch out.sketch threads\n"
                    "  ref_cmd_driver sub ref.sketch qry.sketch out.sketch threads\n  ref_cmd_driver merge list out.sketch threads\n");
    return 2;
}
