#define NS pk
#include "kernel.inc"
