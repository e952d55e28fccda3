// stub: under the CPU SIMT emulation (tests only) everything comes from hip_emu.h, force-included.
#pragma once
