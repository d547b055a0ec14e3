#include "iter_trace.h"
extern "C" { KfposIterTrace kfpos_iter_trace_cur; void kfpos_iter_trace_done() { kfpos_iter_trace_flush(); } }
#include "../../tests/emu/kfpos_emu.cpp"
