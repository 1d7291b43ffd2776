// `fmsi` command-line entry for the MI355X query path.  Only the sub-commands on the hot path are
// provided (`index`, `query`, `lookup`, `export`; reference dispatcher: src/main.cpp:668-699); everything else of the
// reference's CLI (index construction, set operations, export, ...) is out of scope of this build and
// says so instead of silently doing something different.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../include/fmsi_hip.h"

#define FMSI_AMD_VERSION "0.4.0-mi355x.1"

static int usage() {
    std::cerr << std::endl;
    std::cerr << "Program: FMSI (MI355X query path) - k-mer membership queries on masked-superstring indexes." << std::endl;
    std::cerr << "Version: " << FMSI_AMD_VERSION << std::endl << std::endl;
    std::cerr << "Usage:   fmsi <command> [options]" << std::endl << std::endl;
    std::cerr << "Command:" << std::endl;
    std::cerr << "    index   - Creates a BWT based index of the given masked superstring (suffix sort on the GPU)." << std::endl;
    std::cerr << "    query   - Queries k-mers against an index (reads the reference's .fmsi.* files)." << std::endl;
    std::cerr << "    lookup  - Return unique hashes of present k-mers." << std::endl;
    std::cerr << "    export  - Export the underlying masked superstring." << std::endl << std::endl;
    std::cerr << "The index files are the reference's .fmsi.* format, in both directions." << std::endl << std::endl;
    return 1;
}

int main(int argc, char* argv[]) {
    if (argc < 2) return usage();
    const std::string op = argv[1];
    if (op == "query" || op == "lookup") {
        // The process ends with the command: what it wrote is flushed, and the teardown of the device context, of gigabytes of HBM and
        // of the page-locked batch buffers is left to the operating system (0.1-0.2 s of a 0.9 s run at human scale) -- unless something
        // in the process writes its output when the process is finalised: rocprofv3 / rocprofiler-sdk and rocTX (trace.h's ranges),
        // sanitizers, gcov.  Then the command tears down what it built (and waits for a dictionary still being built beside the queries)
        // and the ordinary exit runs the atexit handlers and library destructors.
        bool full_exit = false;
        for (const char* v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "LD_PRELOAD", "ASAN_OPTIONS", "TSAN_OPTIONS",
                              "UBSAN_OPTIONS", "GCOV_PREFIX", "FMSI_HIP_CLI_FULL_EXIT"}) {
            const char* e = std::getenv(v);
            if (e && *e) full_exit = true;
        }
        if (!full_exit) setenv("FMSI_HIP_CLI_NO_TEARDOWN", "1", 0);
        const int rc = fmsi_cli_query(argc - 1, argv + 1, op == "lookup" ? 1 : 0);
        std::cout.flush();
        std::cerr.flush();
        std::fflush(nullptr);
        if (full_exit && !std::getenv("FMSI_HIP_CLI_NO_TEARDOWN")) std::exit(rc);
        _exit(rc);
    }
    if (op == "index") return fmsi_cli_index(argc - 1, argv + 1);
    if (op == "export") return fmsi_cli_export(argc - 1, argv + 1);
    if (op == "-v") {
        std::cout << FMSI_AMD_VERSION << std::endl;
        return 0;
    }
    if (op == "-h") {
        usage();
        return 0;
    }
    if (op == "merge" || op == "compact" || op == "normalize" || op == "clean" ||
        op == "union" || op == "inter" || op == "diff" || op == "symdiff") {
        std::cerr << "ERROR: `fmsi " << op << "` is not part of the MI355X query path; use the reference build for it." << std::endl;
        return 1;
    }
    return usage();
}
