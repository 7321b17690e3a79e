{-# LANGUAGE ForeignFunctionInterface #-}
-- | Binding of libbpgpu.so (include/bpgpu.h) for sdiehl/bulletproofs v1.1.0.
--
-- NOT COMPILED in this repository: the build image has no GHC (DESIGN.md §3).  This is the source a
-- maintainer adds to the reference package (INTEGRATION.md): the exported functions have the types of
-- the reference's own `Bulletproofs.Utils.sumExps`, `addTwoMulP`, `commit`,
-- `Bulletproofs.MultiRangeProof.verifyProof` and `generateProof`, and call the engine instead of
-- evaluating `mul`/`<>` on the CPU.
module Bulletproofs.Gpu
  ( sumExps, addTwoMulP, commit
  , verifyProofs, verifyProof
  , generateProof
  ) where

import Protolude hiding (hash)
import qualified Data.ByteString as BS
import qualified Data.ByteString.Internal as BI
import qualified Data.ByteString.Unsafe as BU
import Data.Bits (shiftL, shiftR, (.|.))
import Foreign
import Foreign.C.String (CString)
import Foreign.C.Types
import System.IO.Unsafe (unsafePerformIO)
import Control.Monad.Random (MonadRandom, getRandomR)
import Control.Monad.Trans.Except (ExceptT, throwE)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr, _r)
import Data.Curve.Weierstrass (Point(..))
import Data.Field.Galois (fromP, toP)

import Bulletproofs.RangeProof.Internal (RangeProof(..), RangeProofError(..))
import Bulletproofs.InnerProductProof.Internal (InnerProductProof(..))
import Bulletproofs.Utils (log2Ceil)

data BpCtx

foreign import ccall safe "bp_ctx_create"        c_ctx_create   :: Ptr (Ptr BpCtx) -> CInt -> IO CInt
foreign import ccall safe "bp_generators_derive" c_gens_derive  :: Ptr BpCtx -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
-- generator CAFs from a file written by benchmarks/gen_tables.py / bp_generators_save (checked on load), and the
-- device's decimal `show` digits (80-byte records) for a transcript kept in Haskell
foreign import ccall safe "bp_generators_load"   c_gens_load    :: Ptr BpCtx -> CString -> IO CInt
foreign import ccall safe "bp_generators_save"   c_gens_save    :: Ptr BpCtx -> CString -> IO CInt
foreign import ccall safe "bp_show_numbers"      c_show_numbers :: Ptr BpCtx -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_msm"               c_msm          :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_add_two_mul"       c_add_two_mul  :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_upper_bound_bits"  c_ub_bits      :: Ptr Word8 -> CSize -> Ptr Word32 -> IO CInt
foreign import ccall safe "bp_rangeproof_prove"  c_prove
  :: Ptr BpCtx -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> CInt -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_rangeproof_verify_batch" c_verify_batch
  :: Ptr BpCtx -> Word32 -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt

-- | One context per process, like the reference's CAF generator lists (Utils.hs:14-25).  Created before
-- anything else touches HIP so that the library's GPU_MAX_HW_QUEUES setting takes effect.
{-# NOINLINE ctx #-}
ctx :: Ptr BpCtx
ctx = unsafePerformIO $ alloca $ \pp -> do
  rc <- c_ctx_create pp 0
  when (rc /= 0) $ panic "bp_ctx_create failed"
  c <- peek pp
  rc' <- c_gens_derive c 1024 nullPtr nullPtr nullPtr   -- h, take 1024 gs, take 1024 hs on the device
  when (rc' /= 0) $ panic "bp_generators_derive failed"
  pure c

---------------------------------------------------------------------------------------------------
-- Marshalling: Fr <-> 32 bytes little-endian, PA <-> 64 bytes x || y little-endian, O <-> 64 zero bytes
---------------------------------------------------------------------------------------------------
le32 :: Integer -> BS.ByteString
le32 v = BS.pack [ fromIntegral (v `shiftR` (8 * i)) | i <- [0 .. 31] ]

unLe :: BS.ByteString -> Integer
unLe = BS.foldr (\b acc -> (acc `shiftL` 8) .|. fromIntegral b) 0

encFr :: Fr -> BS.ByteString
encFr = le32 . fromP

encPt :: PA -> BS.ByteString
encPt O       = BS.replicate 64 0
encPt (A x y) = le32 (fromP x) <> le32 (fromP y)

decPt :: BS.ByteString -> PA
decPt bs
  | BS.all (== 0) bs = O
  | otherwise        = A (toP (unLe (BS.take 32 bs))) (toP (unLe (BS.drop 32 bs)))

withBS :: BS.ByteString -> (Ptr Word8 -> IO a) -> IO a
withBS bs k = BU.unsafeUseAsCString bs (k . castPtr)

check :: Text -> CInt -> IO ()
check what rc = when (rc /= 0) $ panic (what <> " failed: " <> show rc)

---------------------------------------------------------------------------------------------------
-- Utils.hs:72-85
---------------------------------------------------------------------------------------------------
-- | `sumExps` (Utils.hs:76-80): truncates to the shorter list, [] -> O.
sumExps :: [Fr] -> [PA] -> PA
sumExps es ps = unsafePerformIO $ do
  let n = min (length es) (length ps)
  out <- BI.create 64 $ \po ->
    withBS (BS.concat (encFr <$> take n es)) $ \pe ->
    withBS (BS.concat (encPt <$> take n ps)) $ \pp ->
      c_msm ctx pe pp (fromIntegral n) po >>= check "bp_msm"
  pure (decPt out)

-- | `addTwoMulP` (Utils.hs:72-73).
addTwoMulP :: Fr -> PA -> Fr -> PA -> PA
addTwoMulP e0 p0 e1 p1 = unsafePerformIO $ do
  out <- BI.create 64 $ \po ->
    withBS (encFr e0) $ \a -> withBS (encPt p0) $ \b -> withBS (encFr e1) $ \c -> withBS (encPt p1) $ \d ->
      c_add_two_mul ctx a b c d 1 po >>= check "bp_add_two_mul"
  pure (decPt out)

-- | `commit x r = addTwoMulP x gen r h` is expressed through the context's generators by the caller
-- (Utils.hs:84-85); kept here with explicit points so that the module has no second copy of `h`.
commit :: PA -> PA -> Fr -> Fr -> PA
commit g h x r = addTwoMulP x g r h

---------------------------------------------------------------------------------------------------
-- MultiRangeProof/Verifier.hs:19-39 for a batch: `map (\(vs, p) -> verifyProof ub vs p)`
---------------------------------------------------------------------------------------------------
-- | All proofs must share the upper bound and the (padded) number of commitments.
verifyProofs :: Integer -> [([PA], RangeProof Fr PA)] -> [Bool]
verifyProofs _ [] = []
verifyProofs upperBound batch = unsafePerformIO $ do
  let n      = floor (logBase 2 (fromIntegral upperBound :: Double)) :: Int   -- logBase2, Utils.hs:93-94
      m      = 2 ^ log2Ceil (length (fst (unsafeHead batch)))                 -- padding with O, Verifier.hs:37-38
      b      = length batch
      proofs = snd <$> batch
      row f  = BS.concat (f <$> proofs)
      ipp    = productProof
      k      = length (lCommits (ipp (unsafeHead proofs)))
      scalars = BS.concat [ row (encFr . tBlinding), row (encFr . mu), row (encFr . t)
                          , row (encFr . l . ipp), row (encFr . r . ipp) ]
      points  = BS.concat $
                  [ row (encPt . aCommit), row (encPt . sCommit), row (encPt . t1Commit), row (encPt . t2Commit) ]
                  <> [ row (\p -> encPt (lCommits (ipp p) `unsafeIndex` i)) | i <- [0 .. k - 1] ]
                  <> [ row (\p -> encPt (rCommits (ipp p) `unsafeIndex` i)) | i <- [0 .. k - 1] ]
                  <> [ BS.concat [ encPt (fromMaybe O (atMay vs j)) | (vs, _) <- batch ] | j <- [0 .. m - 1] ]
  out <- BI.create b $ \po -> withBS scalars $ \ps -> withBS points $ \pp ->
    c_verify_batch ctx (fromIntegral n) (fromIntegral m) (fromIntegral b) ps pp po
      >>= check "bp_rangeproof_verify_batch"
  pure ((/= 0) <$> BS.unpack out)
  where
    unsafeHead = fromMaybe (panic "empty") . head
    unsafeIndex xs i = fromMaybe (panic "index") (atMay xs i)

verifyProof :: Integer -> [PA] -> RangeProof Fr PA -> Bool
verifyProof ub vs p = and (verifyProofs ub [(vs, p)])

---------------------------------------------------------------------------------------------------
-- MultiRangeProof/Prover.hs:19-123: the MonadRandom draws stay in Haskell, in the reference's order
---------------------------------------------------------------------------------------------------
generateProof :: MonadRandom m => Integer -> [(Fr, Fr)] -> ExceptT (RangeProofError Fr) m (RangeProof Fr PA)
generateProof upperBound vsAndBlindings = do
  let ubBytes = BS.pack (takeWhile (const True) [ fromIntegral (upperBound `shiftR` (8 * i)) | i <- [0 .. 39] ])
      (rcB, nBits) = unsafePerformIO $ alloca $ \pn -> withBS ubBytes $ \pu -> do
        rc <- c_ub_bits pu (fromIntegral (BS.length ubBytes)) pn
        nb <- peek pn
        pure (rc, fromIntegral nb :: Int)
  when (rcB == -6) $ throwE (UpperBoundTooLarge upperBound)
  when (rcB == -7) $ throwE (NNotPowerOf2 upperBound)
  let mRaw = length vsAndBlindings
      m2   = 2 ^ log2Ceil mRaw
      nm   = nBits * m2
      k    = log2Ceil nm
      draw hi = fromInteger <$> lift (getRandomR (1, hi)) :: ExceptT (RangeProofError Fr) m Fr
  sL <- replicateM nm (draw (2 ^ nm - 1))                 -- chooseBlindingVectors, Utils.hs:135-139
  sR <- replicateM nm (draw (2 ^ nm - 1))
  [aB, sB, t1B, t2B] <- replicateM 4 (draw (fromIntegral _r - 1))   -- Prover.hs:69-70,81-82
  let cat = BS.concat . fmap encFr
      (rc, scalarsOut, pointsOut) = unsafePerformIO $ do
        sc  <- BI.mallocByteString (5 * 32)
        pts <- BI.mallocByteString ((4 + 2 * k) * 64)
        rc' <- withForeignPtr sc $ \psc -> withForeignPtr pts $ \ppt ->
          withBS (cat (fst <$> vsAndBlindings)) $ \pv -> withBS (cat (snd <$> vsAndBlindings)) $ \pg ->
          withBS (cat sL) $ \psl -> withBS (cat sR) $ \psr ->
          withBS (encFr aB) $ \pa -> withBS (encFr sB) $ \pr -> withBS (encFr t1B) $ \p1 -> withBS (encFr t2B) $ \p2 ->
            c_prove ctx (fromIntegral nBits) (fromIntegral mRaw) pv pg psl psr pa pr p1 p2 1 psc ppt
        pure (rc', BI.fromForeignPtr sc 0 (5 * 32), BI.fromForeignPtr pts 0 ((4 + 2 * k) * 64))
  when (rc == -5) $ throwE (ValuesNotInRange (fst <$> vsAndBlindings))
  when (rc /= 0) $ panic ("bp_rangeproof_prove failed: " <> show rc)
  let sAt i = toP (unLe (BS.take 32 (BS.drop (32 * i) scalarsOut))) :: Fr
      pAt i = decPt (BS.take 64 (BS.drop (64 * i) pointsOut))
  pure RangeProof
    { tBlinding = sAt 0, mu = sAt 1, t = sAt 2
    , aCommit = pAt 0, sCommit = pAt 1, t1Commit = pAt 2, t2Commit = pAt 3
    , productProof = InnerProductProof
        { lCommits = pAt <$> [4 .. 3 + k], rCommits = pAt <$> [4 + k .. 3 + 2 * k], l = sAt 3, r = sAt 4 } }
