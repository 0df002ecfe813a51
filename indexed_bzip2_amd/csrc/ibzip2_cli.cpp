/**
 * ibzip2-mi355x -- command line front end over the C ABI (include/mi355x_bz2.h).
 *
 * Mirrors the options and the behaviour of the reference tool `ibzip2` (src/tools/ibzip2.cpp:172-480): -c -d -f -i -o
 * -k -t -p -P -h -q -v -V -l -L --buffer-size, the same rules for the output file name, for refusing to overwrite,
 * for where the offset lists go (file / stdout / stderr) and the same two text formats:
 *   -l : one compressed bit offset per line                                   (dumpOffsets, ibzip2.cpp:68-80)
 *   -L : "<compressed bit offset>,<decoded byte offset>" per line             (dumpOffsets, ibzip2.cpp:83-93)
 * The reference parses its command line with cxxopts (an un-vendored submodule); this parser is our own.
 *
 * Differences: -P is the number of blocks kept in flight per GPU batch and defaults to 0 = automatic (512): there is no
 * serial CPU decoder behind -P 1, which would mean one block per GPU launch here.  What the reference's serial decoder
 * checks and its parallel one does not -- the combined CRC of every stream -- is checked with -t (and with -P 1).
 * Standard input is read completely into memory first (the GPU decodes from a resident copy anyway).
 */
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mi355x_bz2.h"

namespace
{
struct Options
{
    bool toStdout{ false }, decompress{ false }, force{ false }, keep{ false }, test{ false }, help{ false };
    bool quiet{ false }, version{ false };
    int verbose{ 0 };
    bool listCompressed{ false }, listOffsets{ false };
    std::string input, output, listCompressedPath, listOffsetsPath;
    bool hasOutput{ false };
    unsigned finderParallelism{ 1 }, decoderParallelism{ 0 }, bufferSize{ 0 };
    int device{ -1 };
};

void
printHelp()
{
    std::cout <<
        "A bzip2 decompressor tool based on the MI355X block decoder (drop-in for ibzip2)\n"
        "Usage:\n"
        "  ibzip2-mi355x [OPTION...] [input file]\n\n"
        " Decompression options:\n"
        "  -c, --stdout                  Output to standard output. This is the default, when reading from standard input.\n"
        "  -d, --decompress              Force decompression. Only for compatibility. No compression supported anyways.\n"
        "  -f, --force                   Force overwriting existing output files.\n"
        "  -i, --input arg               Input file. If none is given, data is read from standard input.\n"
        "  -o, --output arg              Output file. If none is given, use the input file name with '.bz2' stripped or\n"
        "                                '<input file>.out'.\n"
        "  -k, --keep                    Keep (do not delete) input file. Only for compatibility.\n"
        "  -t, --test                    Test compressed file integrity.\n"
        "  -p, --block-finder-parallelism arg\n"
        "                                Threads of the host block finder (0 = automatic). (default: 1)\n"
        "  -P, --decoder-parallelism arg Blocks kept in flight per GPU batch (0 = automatic). (default: 0; the\n"
        "                                reference defaults to 1 = its serial CPU decoder, which does not exist here)\n"
        "      --device arg              GPU to use (default: current device)\n\n"
        " Output options:\n"
        "  -h, --help                    Print this help message.\n"
        "  -q, --quiet                   Suppress noncritical error messages.\n"
        "  -v, --verbose                 Be verbose. A second -v (or shorthand -vv) gives even more verbosity.\n"
        "  -V, --version                 Display software version.\n"
        "  -l, --list-compressed-offsets [arg]\n"
        "                                List only the bzip2 block offsets given in bits one per line to the specified\n"
        "                                output file. If no file is given, it will print to stdout or to stderr if the\n"
        "                                decoded data is already written to stdout.\n"
        "  -L, --list-offsets [arg]      List bzip2 block offsets in bits and also the corresponding offsets in the\n"
        "                                decoded data at the beginning of each block in bytes as comma separated pairs\n"
        "                                per line '<encoded bits>,<decoded bytes>'.\n\n"
        " Advanced options:\n"
        "      --buffer-size arg         Controls the output buffer size. By default, the decoded data is written in\n"
        "                                one pass per block. (default: 0)\n\n"
        "Examples:\n\n"
        "Decompress a file:\n  ibzip2-mi355x -d file.bz2\n\n"
        "Decompress a file with 256 blocks per GPU batch:\n  ibzip2-mi355x -d -P 256 file.bz2\n\n"
        "Find and list the bzip2 block offsets to be used for another tool:\n"
        "  ibzip2-mi355x -l blockoffsets.dat -- file.bz2\n\n"
        "List block offsets in both the compressed as well as the decompressed data:\n"
        "  ibzip2-mi355x -L blockoffsets.dat file.bz2 > /dev/null\n";
}

bool
fileExists( const std::string& path )
{
    struct stat st{};
    return ::stat( path.c_str(), &st ) == 0;
}

bool
endsWithNoCase( const std::string& s, const std::string& suffix )
{
    if ( s.size() < suffix.size() ) return false;
    for ( size_t i = 0; i < suffix.size(); ++i ) {
        if ( std::tolower( (unsigned char)s[s.size() - suffix.size() + i] ) != std::tolower( (unsigned char)suffix[i] ) ) return false;
    }
    return true;
}

bool
parseUnsigned( const char* text, unsigned& value )
{
    if ( text == nullptr || *text == '\0' ) return false;
    char* end = nullptr;
    errno = 0;
    const unsigned long v = std::strtoul( text, &end, 10 );
    if ( errno != 0 || *end != '\0' || v > 0xFFFFFFFFul ) return false;
    value = (unsigned)v;
    return true;
}

/** Returns 0 on success.  An option with an optional value (-l, -L) takes the next argument unless that starts with '-'
 * or is the last argument while no input has been seen (then it is the positional input file, the ambiguity the
 * reference warns about, ibzip2.cpp:175-178, resolved the way its examples are written: "-l file -- input"). */
int
parseArguments( int argc, char** argv, Options& o )
{
    std::vector<std::string> positional;
    bool onlyPositional = false;
    const auto optionalValue = [&] ( int& i, std::string& path ) {
        if ( i + 1 < argc && argv[i + 1][0] != '-' && !( i + 2 == argc && o.input.empty() && positional.empty() ) ) {
            path = argv[++i];
        }
    };
    const auto requiredValue = [&] ( int& i, const std::string& name, std::string& out ) -> bool {
        if ( i + 1 >= argc ) {
            std::cerr << "Option '" << name << "' is missing an argument\n";
            return false;
        }
        out = argv[++i];
        return true;
    };
    for ( int i = 1; i < argc; ++i ) {
        const std::string a = argv[i];
        if ( onlyPositional || a.empty() || a[0] != '-' || a == "-" ) {
            positional.push_back( a );
            continue;
        }
        if ( a == "--" ) { onlyPositional = true; continue; }
        std::string value;
        if ( a.rfind( "--", 0 ) == 0 ) {
            std::string name = a.substr( 2 ), inlineValue;
            bool hasInline = false;
            const auto eq = name.find( '=' );
            if ( eq != std::string::npos ) {
                inlineValue = name.substr( eq + 1 );
                name = name.substr( 0, eq );
                hasInline = true;
            }
            const auto need = [&] ( std::string& out ) -> bool {
                if ( hasInline ) { out = inlineValue; return true; }
                return requiredValue( i, "--" + name, out );
            };
            if ( name == "stdout" ) o.toStdout = true;
            else if ( name == "decompress" ) o.decompress = true;
            else if ( name == "force" ) o.force = true;
            else if ( name == "keep" ) o.keep = true;
            else if ( name == "test" ) o.test = true;
            else if ( name == "help" ) o.help = true;
            else if ( name == "quiet" ) o.quiet = true;
            else if ( name == "verbose" ) ++o.verbose;
            else if ( name == "version" ) o.version = true;
            else if ( name == "input" ) { if ( !need( o.input ) ) return 1; }
            else if ( name == "output" ) { if ( !need( o.output ) ) return 1; o.hasOutput = true; }
            else if ( name == "list-compressed-offsets" ) {
                o.listCompressed = true;
                if ( hasInline ) o.listCompressedPath = inlineValue; else optionalValue( i, o.listCompressedPath );
            } else if ( name == "list-offsets" ) {
                o.listOffsets = true;
                if ( hasInline ) o.listOffsetsPath = inlineValue; else optionalValue( i, o.listOffsetsPath );
            } else if ( name == "block-finder-parallelism" ) {
                if ( !need( value ) || !parseUnsigned( value.c_str(), o.finderParallelism ) ) { std::cerr << "Bad value for --" << name << "\n"; return 1; }
            } else if ( name == "decoder-parallelism" ) {
                if ( !need( value ) || !parseUnsigned( value.c_str(), o.decoderParallelism ) ) { std::cerr << "Bad value for --" << name << "\n"; return 1; }
            } else if ( name == "buffer-size" ) {
                if ( !need( value ) || !parseUnsigned( value.c_str(), o.bufferSize ) ) { std::cerr << "Bad value for --" << name << "\n"; return 1; }
            } else if ( name == "device" ) {
                unsigned d = 0;
                if ( !need( value ) || !parseUnsigned( value.c_str(), d ) ) { std::cerr << "Bad value for --device\n"; return 1; }
                o.device = (int)d;
            } else {
                std::cerr << "Option '" << name << "' does not exist\n";
                return 1;
            }
            continue;
        }
        /* bundle of short options, e.g. -dvv or -P0 */
        for ( size_t k = 1; k < a.size(); ++k ) {
            const char c = a[k];
            const auto rest = [&] ( std::string& out ) -> bool {   /* value glued to the option or the next argument */
                if ( k + 1 < a.size() ) { out = a.substr( k + 1 ); k = a.size(); return true; }
                return requiredValue( i, std::string( "-" ) + c, out );
            };
            switch ( c ) {
            case 'c': o.toStdout = true; break;
            case 'd': o.decompress = true; break;
            case 'f': o.force = true; break;
            case 'k': o.keep = true; break;
            case 't': o.test = true; break;
            case 'h': o.help = true; break;
            case 'q': o.quiet = true; break;
            case 'v': ++o.verbose; break;
            case 'V': o.version = true; break;
            case 'i': if ( !rest( o.input ) ) return 1; break;
            case 'o': if ( !rest( o.output ) ) return 1; o.hasOutput = true; break;
            case 'p': if ( !rest( value ) || !parseUnsigned( value.c_str(), o.finderParallelism ) ) { std::cerr << "Bad value for -p\n"; return 1; } break;
            case 'P': if ( !rest( value ) || !parseUnsigned( value.c_str(), o.decoderParallelism ) ) { std::cerr << "Bad value for -P\n"; return 1; } break;
            case 'l':
                o.listCompressed = true;
                if ( k + 1 < a.size() ) { o.listCompressedPath = a.substr( k + 1 ); k = a.size(); } else optionalValue( i, o.listCompressedPath );
                break;
            case 'L':
                o.listOffsets = true;
                if ( k + 1 < a.size() ) { o.listOffsetsPath = a.substr( k + 1 ); k = a.size(); } else optionalValue( i, o.listOffsetsPath );
                break;
            default:
                std::cerr << "Option '" << c << "' does not exist\n";
                return 1;
            }
        }
    }
    if ( positional.size() + ( o.input.empty() ? 0 : 1 ) > 1 ) {
        std::cerr << "One or none bzip2 filename to decompress must be specified!\n";
        return 1;
    }
    if ( !positional.empty() ) o.input = positional.front();
    return 0;
}

bool
stdinHasInput()
{
    return !::isatty( STDIN_FILENO );
}

/** Whole input in memory: a mapping for files, a vector for standard input. */
struct Input
{
    const uint8_t* data{ nullptr };
    uint64_t size{ 0 };
    std::vector<uint8_t> owned;
    void* mapping{ nullptr };
    ~Input() { if ( mapping != nullptr ) ::munmap( mapping, size ); }

    bool
    open( const std::string& path )
    {
        if ( path.empty() ) {
            uint8_t buffer[1 << 16];
            for ( ;; ) {
                const ssize_t n = ::read( STDIN_FILENO, buffer, sizeof( buffer ) );
                if ( n < 0 && errno == EINTR ) continue;
                if ( n <= 0 ) break;
                owned.insert( owned.end(), buffer, buffer + n );
            }
            data = owned.data();
            size = owned.size();
            return true;
        }
        const int fd = ::open( path.c_str(), O_RDONLY );
        if ( fd < 0 ) return false;
        struct stat st{};
        if ( ::fstat( fd, &st ) != 0 ) { ::close( fd ); return false; }
        size = (uint64_t)st.st_size;
        if ( size > 0 ) {
            mapping = ::mmap( nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0 );
            if ( mapping == MAP_FAILED ) { mapping = nullptr; ::close( fd ); return false; }
            data = static_cast<const uint8_t*>( mapping );
        }
        ::close( fd );
        return true;
    }
};

constexpr uint64_t MAGIC_BLOCK = 0x314159265359ull;
constexpr uint64_t MAGIC_EOS = 0x177245385090ull;

/** checkOffsets, ibzip2.cpp:37-65: every listed offset must point at one of the two 48-bit magics. */
bool
checkOffsets( const Input& in, const std::vector<uint64_t>& offsets )
{
    for ( const auto offset : offsets ) {
        uint64_t magic = 0;
        bool inside = offset + 48 <= in.size * 8;
        for ( uint64_t b = offset; inside && b < offset + 48; ++b ) {
            magic = ( magic << 1 ) | ( ( in.data[b >> 3] >> ( 7 - ( b & 7 ) ) ) & 1u );
        }
        if ( !inside || ( magic != MAGIC_BLOCK && magic != MAGIC_EOS ) ) {
            std::cerr << "Magic bytes " << std::hex << magic << std::dec << " at offset " << ( offset / 8 ) << " B "
                      << ( offset % 8 ) << "b do not match bzip2 magic bytes!\n";
            return false;
        }
    }
    return true;
}

void
dumpOffsets( std::ostream& out, const std::vector<uint64_t>& offsets )
{
    if ( !out.good() ) return;
    for ( const auto offset : offsets ) out << offset << "\n";
}

void
dumpOffsets( std::ostream& out, const std::vector<uint64_t>& bits, const std::vector<uint64_t>& bytes )
{
    if ( !out.good() ) return;
    for ( size_t i = 0; i < bits.size(); ++i ) out << bits[i] << "," << bytes[i] << "\n";
}

/** findCompressedBlocks, ibzip2.cpp:96-139: both magics, sorted, no decoding (runs without a GPU). */
int
findCompressedBlocks( const Options& o )
{
    Input in;
    if ( !in.open( o.input ) ) {
        std::cerr << "Could not open '" << o.input << "'\n";
        return 1;
    }
    std::vector<uint64_t> offsets;
    for ( const auto magic : { MAGIC_BLOCK, MAGIC_EOS } ) {
        const uint64_t n = mi355x_bz2_find_magic( in.data, in.size, magic, nullptr, 0, o.finderParallelism );
        std::vector<uint64_t> found( n );
        mi355x_bz2_find_magic( in.data, in.size, magic, found.data(), n, o.finderParallelism );
        offsets.insert( offsets.end(), found.begin(), found.end() );
    }
    std::sort( offsets.begin(), offsets.end() );
    if ( o.test && !checkOffsets( in, offsets ) ) return 1;
    if ( o.listCompressedPath.empty() ) {
        dumpOffsets( std::cout, offsets );
    } else {
        std::ofstream file( o.listCompressedPath );
        dumpOffsets( file, offsets );
    }
    if ( o.verbose > 0 ) std::cout << "Found " << offsets.size() << " blocks\n";
    return 0;
}
}  // namespace

int
main( int argc, char** argv )
{
    /* the reader drives two decoder contexts with four HIP streams each: give every stream its own hardware queue
     * (read by the HIP runtime when it starts; an existing setting wins) */
    ::setenv( "GPU_MAX_HW_QUEUES", "16", 0 );
    Options o;
    if ( parseArguments( argc, argv, o ) != 0 ) return 1;

    if ( o.help ) {
        printHelp();
        return 0;
    }
    if ( o.version ) {
        std::cout << "ibzip2-mi355x, CLI to the MI355X bzip2 block decoder (C ABI version " << mi355x_bz2_abi_version()
                  << "), option-compatible with ibzip2 of indexed-bzip2 1.7.0.\n";
        return 0;
    }
    if ( !stdinHasInput() && o.input.empty() ) {
        std::cerr << "Either stdin must have input, e.g., by piping to it, or an input file must be specified!\n";
        return 1;
    }

    /* output file name rules, ibzip2.cpp:316-331 */
    std::string outputPath = o.output;
    if ( !o.toStdout && outputPath.empty() && !o.input.empty() ) {
        if ( endsWithNoCase( o.input, ".bz2" ) ) {
            outputPath = o.input.substr( 0, o.input.size() - 4 );
        } else {
            outputPath = o.input + ".out";
            if ( !o.quiet ) std::cerr << "Could not deduce output file name. Will write to '" << outputPath << "'\n";
        }
    }
    if ( o.verbose > 0 ) {
        const auto show = [] ( const std::string& v, bool given ) { return v.empty() ? ( given ? "<stdout>" : "<none>" ) : v.c_str(); };
        std::cerr << "file path for input: " << show( o.input, false ) << "\n"
                  << "file path for output: " << show( o.output, o.hasOutput ) << "\n"
                  << "file path for list-compressed-offsets: " << show( o.listCompressedPath, o.listCompressed ) << "\n"
                  << "file path for list-offsets: " << show( o.listOffsetsPath, o.listOffsets ) << "\n";
    }
    if ( o.decompress && outputPath != "/dev/null" && !outputPath.empty() && fileExists( outputPath ) && !o.force ) {
        std::cerr << "Output file '" << outputPath << "' already exists! Use --force to overwrite.\n";
        return 1;
    }
    if ( !o.listOffsetsPath.empty() && fileExists( o.listOffsetsPath ) && !o.force ) {
        std::cerr << "Output file for offsets'" << o.listOffsetsPath << "' for offsets already exists! Use --force to overwrite.\n";
        return 1;
    }
    if ( !o.listCompressedPath.empty() && fileExists( o.listCompressedPath ) && !o.force ) {
        std::cerr << "Output file compressed offsets '" << o.listCompressedPath
                  << "' for offsets already exists! Use --force to overwrite.\n";
        return 1;
    }

    if ( o.decompress || o.listOffsets ) {
        if ( o.verbose > 0 ) {
            std::cerr << "Decompress " << o.input << " -> " << outputPath << " with " << o.decoderParallelism
                      << " blocks per GPU batch\n";
        }
        Input in;
        if ( !in.open( o.input ) ) {
            std::cerr << "Could not open '" << o.input << "'\n";
            return 1;
        }
        /* the reference's default (serial) reader starts with readBzip2Header (bzip2.hpp:114-142) and throws on
         * anything else; its parallel reader would silently produce nothing for an input without a block magic */
        if ( mi355x_bz2_read_stream_header( in.data, in.size, 0 ) == 0 ) {
            std::cerr << "Decoding failed: " << mi355x_bz2_status_string( MI355X_BZ2_ERR_STREAM_HEADER ) << "\n";
            return 1;
        }
        int outFd = -1;
        bool writingToStdout = false;
        if ( o.decompress ) {
            if ( outputPath.empty() ) {
                outFd = STDOUT_FILENO;
                writingToStdout = true;
            } else {
                outFd = ::open( outputPath.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644 );
                if ( outFd < 0 ) {
                    std::cerr << "Could not open output file '" << outputPath << "' for writing!\n";
                    return 1;
                }
            }
        }

        const auto tOpen = std::chrono::steady_clock::now();
        mi355x_bz2_reader* reader = nullptr;
        int rc = mi355x_bz2_reader_open_memory( in.data, in.size, o.decoderParallelism, o.device, &reader );
        if ( rc != MI355X_BZ2_OK ) {
            std::cerr << "Could not open the bzip2 stream: " << mi355x_bz2_status_string( rc ) << "\n";
            if ( outFd >= 0 && !writingToStdout ) ::close( outFd );
            return 1;
        }
        if ( o.test ) (void)mi355x_bz2_reader_set_verify_stream_crc( reader, 1 );
        const auto fail = [&] ( int status ) {
            const char* detail = mi355x_bz2_reader_last_error( reader );
            std::cerr << "Decoding failed: " << mi355x_bz2_status_string( status );
            if ( detail != nullptr && detail[0] != '\0' ) std::cerr << " (" << detail << ")";
            std::cerr << "\n";
            mi355x_bz2_reader_close( reader );
            if ( outFd >= 0 && !writingToStdout ) ::close( outFd );
            return 1;
        };

        const auto tRead = std::chrono::steady_clock::now();
        uint64_t written = 0;
        if ( o.bufferSize > 0 ) {
            std::vector<char> buffer( o.bufferSize );
            do {
                uint64_t got = 0;
                rc = mi355x_bz2_reader_read( reader, -1, buffer.data(), buffer.size(), &got );
                if ( rc != MI355X_BZ2_OK ) return fail( rc );
                if ( outFd >= 0 ) {
                    uint64_t done = 0;
                    while ( done < got ) {
                        const ssize_t n = ::write( outFd, buffer.data() + done, got - done );
                        if ( n < 0 && errno == EINTR ) continue;
                        if ( n <= 0 ) {
                            std::cerr << "Could not write all the decoded data to the specified output!\n";
                            break;
                        }
                        done += (uint64_t)n;
                    }
                }
                written += got;
            } while ( !mi355x_bz2_reader_eof( reader ) );
        } else {
            rc = mi355x_bz2_reader_read( reader, outFd, nullptr, UINT64_MAX, &written );
            if ( rc != MI355X_BZ2_OK ) return fail( rc );
        }
        if ( outFd >= 0 && !writingToStdout ) ::close( outFd );
        const auto tDone = std::chrono::steady_clock::now();

        std::ostream& out = writingToStdout ? std::cerr : std::cout;

        uint64_t count = 0;
        rc = mi355x_bz2_reader_block_offsets( reader, nullptr, nullptr, 0, &count );
        if ( rc != MI355X_BZ2_OK ) return fail( rc );
        std::vector<uint64_t> bits( count ), bytes( count );
        rc = mi355x_bz2_reader_block_offsets( reader, bits.data(), bytes.data(), count, &count );
        if ( rc != MI355X_BZ2_OK ) return fail( rc );
        if ( o.verbose > 0 ) out << "Found " << count << " blocks\n";

        if ( o.test ) {
            /* the final entry of the map is the end-of-file sentinel, not a magic (BlockMap.hpp:207-230) */
            std::vector<uint64_t> magics( bits.begin(), bits.end() - ( bits.empty() ? 0 : 1 ) );
            if ( !checkOffsets( in, magics ) ) { mi355x_bz2_reader_close( reader ); return 1; }
            uint64_t size = 0;
            if ( !mi355x_bz2_reader_size( reader, &size ) ) {
                std::cerr << "Bzip2 reader size should be available at this point!\n";
                mi355x_bz2_reader_close( reader );
                return 1;
            }
            if ( written != size ) {
                std::cerr << "Wrote less bytes (" << written << " B) than decoded stream is large(" << size << " B)!\n";
                mi355x_bz2_reader_close( reader );
                return 1;
            }
        }
        if ( o.verbose > 1 ) {
            mi355x_bz2_reader_stats st{};
            if ( mi355x_bz2_reader_statistics( reader, &st ) == MI355X_BZ2_OK ) {
                std::cerr << "[statistics] gets " << st.gets << ", cache hits " << st.cache_hits << ", prefetch hits "
                          << st.prefetch_hits << ", on-demand " << st.on_demand_fetches << ", GPU batches " << st.batches
                          << ", blocks decoded " << st.blocks_decoded << ", decode " << st.decode_seconds << " s, wait "
                          << st.wait_seconds << " s\n";
            }
        }
        mi355x_bz2_reader_close( reader );
        if ( o.verbose > 1 ) {
            const auto seconds = [] ( auto a, auto b ) { return std::chrono::duration<double>( b - a ).count(); };
            std::cerr << "[timing] open " << seconds( tOpen, tRead ) << " s, decode " << seconds( tRead, tDone )
                      << " s, close " << seconds( tDone, std::chrono::steady_clock::now() ) << " s\n";
        }

        if ( o.listOffsets ) {
            if ( !o.listOffsetsPath.empty() ) {
                std::ofstream file( o.listOffsetsPath );
                dumpOffsets( file, bits, bytes );
            } else if ( outputPath.empty() ) {
                dumpOffsets( std::cerr, bits, bytes );
            } else {
                dumpOffsets( std::cout, bits, bytes );
            }
        }
        if ( o.listCompressed ) {
            if ( !o.listCompressedPath.empty() ) {
                std::ofstream file( o.listCompressedPath );
                dumpOffsets( file, bits );
            } else if ( outputPath.empty() ) {
                dumpOffsets( std::cerr, bits );
            } else {
                dumpOffsets( std::cout, bits );
            }
        }
        return 0;
    }

    if ( o.listCompressed ) {
        if ( o.verbose > 0 ) std::cerr << "Find block offsets\n";
        return findCompressedBlocks( o );
    }

    std::cerr << "No suitable arguments were given. Please refer to the help!\n\n";
    printHelp();
    return 1;
}
