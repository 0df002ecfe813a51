// TEMPORARY stub so that the library exports every symbol while the reader is being written.
#include "../../include/mi355x_bz2.h"
extern "C" {
int mi355x_bz2_reader_open_path( const char*, uint32_t, int32_t, mi355x_bz2_reader** ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_open_fd( int, uint32_t, int32_t, mi355x_bz2_reader** ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_open_memory( const uint8_t*, uint64_t, uint32_t, int32_t, mi355x_bz2_reader** ) { return MI355X_BZ2_ERR_LOGIC; }
void mi355x_bz2_reader_close( mi355x_bz2_reader* ) {}
const char* mi355x_bz2_reader_last_error( const mi355x_bz2_reader* ) { return ""; }
int mi355x_bz2_reader_read( mi355x_bz2_reader*, int, void*, uint64_t, uint64_t* ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_seek( mi355x_bz2_reader*, int64_t, int, uint64_t* ) { return MI355X_BZ2_ERR_LOGIC; }
uint64_t mi355x_bz2_reader_tell( const mi355x_bz2_reader* ) { return 0; }
int mi355x_bz2_reader_eof( const mi355x_bz2_reader* ) { return 0; }
int mi355x_bz2_reader_closed( const mi355x_bz2_reader* ) { return 1; }
int mi355x_bz2_reader_size( const mi355x_bz2_reader*, uint64_t* ) { return 0; }
uint64_t mi355x_bz2_reader_tell_compressed( const mi355x_bz2_reader* ) { return 0; }
int mi355x_bz2_reader_block_offsets_complete( const mi355x_bz2_reader* ) { return 0; }
int mi355x_bz2_reader_block_offsets( mi355x_bz2_reader*, uint64_t*, uint64_t*, uint64_t, uint64_t* ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_available_block_offsets( const mi355x_bz2_reader*, uint64_t*, uint64_t*, uint64_t, uint64_t* ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_set_block_offsets( mi355x_bz2_reader*, const uint64_t*, const uint64_t*, uint64_t ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_join_threads( mi355x_bz2_reader* ) { return MI355X_BZ2_ERR_LOGIC; }
int mi355x_bz2_reader_statistics( const mi355x_bz2_reader*, mi355x_bz2_reader_stats* ) { return MI355X_BZ2_ERR_LOGIC; }
}
