#define NS ref
#include "kernel.inc"
