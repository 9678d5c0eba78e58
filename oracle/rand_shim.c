/* TEST INFRASTRUCTURE ONLY.  LD_PRELOAD shim that pins libc's rand()/srand()
 * so the UNMODIFIED reference library (oracle/_ref/libfps_ref.so) starts its
 * random-start FPS (farthest_point_sampling.cpp:93-94) at index
 * FAKE_RAND % pn.  Used by tools/oracle/gen_fps_golden.py in a subprocess. */
#include <stdlib.h>
int rand(void) { const char *e = getenv("FAKE_RAND"); return e ? atoi(e) : 0; }
void srand(unsigned int s) { (void)s; }
