// dab/database/dab_database_entities.h -- ONLY the `Subchannel` record the channel decoder reads
// (reference: src/dab/database/dab_database_entities.h:31-46,179-190 and dab_database_types.h). The rest of the
// database (services, components, links) is application-layer state outside the hot path; in the reference tree the
// reference's own header is used.
#pragma once
#include <stdint.h>

typedef uint8_t subchannel_id_t;
typedef uint16_t subchannel_addr_t;
typedef uint16_t subchannel_size_t;
typedef uint8_t eep_protection_level_t;
typedef uint8_t uep_protection_index_t;

enum class EEP_Type : uint8_t { TYPE_A = 0, TYPE_B = 1, UNDEFINED = 0xFF };
enum class FEC_Scheme : uint8_t { NONE = 0, REED_SOLOMON = 1, RFA0 = 2, RFA1 = 3, UNDEFINED = 0xFF };

struct Subchannel {
    subchannel_id_t id;
    subchannel_addr_t start_address = 0;
    subchannel_size_t length = 0;
    bool is_uep = false;
    uep_protection_index_t uep_prot_index = 0;
    eep_protection_level_t eep_prot_level = 0;
    EEP_Type eep_type = EEP_Type::UNDEFINED;
    FEC_Scheme fec_scheme = FEC_Scheme::UNDEFINED;
    bool is_complete = false;
    explicit Subchannel(const subchannel_id_t _id) : id(_id) {}
};
