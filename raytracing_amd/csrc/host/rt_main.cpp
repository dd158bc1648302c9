// `rt` executable: same command line as the reference's src/rt.cpp, pixel loop on the GPU.
#include "../../../include/rt_host.h"
int main(int argc, char **argv) { return rt_main(argc, argv); }
