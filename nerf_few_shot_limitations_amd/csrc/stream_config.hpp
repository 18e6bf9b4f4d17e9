// stream_config.hpp -- geometry of the packed weight stream, shared by the host packer and the kernels
#pragma once
#ifndef NRF_CHUNK_FRAGS
#define NRF_CHUNK_FRAGS 16        // 1-KiB fragments per chunk (one barrier per chunk)
#endif
#ifndef NRF_SLOTS
#define NRF_SLOTS 6               // LDS ring depth in chunks (NRF_CHUNK_FRAGS * NRF_SLOTS KiB); 6 and 8 measure the same (profiles/README.md)
#endif
