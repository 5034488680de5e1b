/* Source-compatibility header: the reference spreads its host API over src/texture.cuh and others;
 * here the whole API lives in path_tracer.h. */
#include "../path_tracer.h"
